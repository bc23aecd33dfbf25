"""CodingCDFNonshared -- the reference's channel iterator over a predicted distribution (bitcoding/coders_helpers.py:31-56):
hands out the CDFOut of channel 0, 1, ... given the channels decoded so far (the RGB scale couples them)."""
import torch

from ..criterion.logistic_mixture import CDFOut  # noqa: F401  (re-exported like the reference module)


class CodingCDFNonshared(object):
    def __init__(self, l, total_C, dmll):
        """l: predicted distribution (N,Kp,H,W); dmll: the DiscretizedMixLogisticLoss of this scale."""
        self.l = l
        self.dmll = dmll
        # bin edges (coders_helpers.py:42-44): one definition for every caller, DiscretizedMixLogisticLoss.coding_targets
        self.targets = dmll.coding_targets(l.device)
        self.total_C = total_C
        self.c_cur = 0

    def get_next_C(self, decoded_x):
        """decoded_x: (N,C,H,W) values decoded so far (only channels < c_cur are read) -> CDFOut of channel c_cur."""
        C_cur = self.dmll.cdf_step_non_shared(self.l, self.targets, self.c_cur, self.total_C, decoded_x)
        self.c_cur += 1
        return C_cur
