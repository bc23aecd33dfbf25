import os


def listdir_paths(p):
    for fn in os.listdir(p):
        yield os.path.join(p, fn)
