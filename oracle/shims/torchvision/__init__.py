from . import utils, transforms  # noqa
