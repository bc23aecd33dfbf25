import time
from contextlib import contextmanager


class TimeAccumulator(object):
    def __init__(self):
        self.times = []

    @contextmanager
    def execute(self):
        t = time.time()
        yield
        self.times.append(time.time() - t)

    def mean_time_spent(self):
        return sum(self.times) / max(len(self.times), 1)
