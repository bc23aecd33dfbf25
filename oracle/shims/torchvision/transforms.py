class Compose(object):
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, x):
        for t in self.ts:
            x = t(x)
        return x


class Lambda(object):
    def __init__(self, f):
        self.f = f

    def __call__(self, x):
        return self.f(x)


class CenterCrop(object):
    def __init__(self, size):
        self.size = (size, size) if isinstance(size, int) else tuple(size)

    def __call__(self, img):
        w, h = img.size
        th, tw = self.size
        l, t = (w - tw) // 2, (h - th) // 2
        return img.crop((l, t, l + tw, t + th))


class RandomCrop(CenterCrop):
    pass


class RandomHorizontalFlip(object):
    def __call__(self, x):
        return x


class ToTensor(object):
    def __call__(self, x):
        raise NotImplementedError
