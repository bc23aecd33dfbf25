"""A set of test images: a directory (every file with an image extension, sorted) or one image file
(reference helpers/testset.py:32-105, helpers/paths.py:36)."""
import os
from functools import total_ordering

import numpy as np

IMG_EXTENSIONS = {'.jpg', '.jpeg', '.png', '.ppm', '.bmp', '.pgm', '.tif'}


def has_image_ext(p):
    return os.path.splitext(p)[1].lower() in IMG_EXTENSIONS


@total_ordering
class Testset(object):
    def __init__(self, root_dir_or_img, max_imgs=None, skip_hidden=False, append_id=None):
        self.root_dir_or_img = root_dir_or_img
        if os.path.isdir(root_dir_or_img):
            root = root_dir_or_img
            self.name = os.path.basename(root.rstrip('/'))
            self.ps = sorted(os.path.join(root, f) for f in os.listdir(root) if has_image_ext(f))
            if skip_hidden:
                self.ps = [p for p in self.ps if not os.path.basename(p).startswith('.')]
            if max_imgs and max_imgs < len(self.ps):
                print('Subsampling to use {} imgs of {}...'.format(max_imgs, self.name))
                idxs = np.linspace(0, len(self.ps) - 1, max_imgs).astype(int)
                self.ps = [self.ps[i] for i in idxs]
            if not self.ps:
                raise ValueError('No images found in {}'.format(root))
            self.id = '{}_{}'.format(self.name, len(self.ps))
            self._str = 'Testset({}): in {}, {} images'.format(self.name, root, len(self.ps))
        else:
            if not os.path.isfile(root_dir_or_img):
                raise FileNotFoundError('Does not exist: {}'.format(root_dir_or_img))
            self.name = os.path.basename(root_dir_or_img)
            self.ps = [root_dir_or_img]
            self.id = root_dir_or_img
            self._str = 'Testset([{}]): 1 image'.format(self.name)
        if append_id:
            self.id += append_id

    def filter_filenames(self, names):
        self.ps = [p for p in self.ps if os.path.splitext(os.path.basename(p))[0] in names]
        if not self.ps:
            raise ValueError('No files after filtering for {}'.format(names))

    def __len__(self):
        return len(self.ps)

    def __str__(self):
        return self._str

    def __eq__(self, other):
        return self.id == other.id

    def __lt__(self, other):
        return self.id < other.id

    def __hash__(self):
        return hash(self.id)
