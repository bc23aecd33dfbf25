"""Encoder / decoder output records, same fields as the reference (modules/net.py:36-43)."""
from collections import namedtuple

EncOut = namedtuple('EncOut', ['bn',    # NCH'W' (eval: equals bn_q, see multiscale_network.Out.append)
                               'bn_q',  # quantized bn, NCH'W'
                               'S',     # NCH'W', long
                               'L',     # int
                               'F'      # NCfH'W', float, before Q
                               ])
DecOut = namedtuple('DecOut', ['F'])
