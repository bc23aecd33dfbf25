"""oracle/ -- CPU restatement of the reference's L3C inference path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import anything from here,
and only as the *checker*.  The product package (`l3c-pytorch_amd/`) never imports `oracle` and has no CPU
fallback: without the HIP library it raises.

Pinning status (see DESIGN.md "Oracle"): the reference ships NO golden vectors or KATs for this path
(SURVEY.md section 4), so every restatement here is pinned against the reference ITSELF run in the build
container (`oracle/ref_import.py` imports the unmodified reference; `oracle/build_ref.py` compiles its own
torchac.cpp into `oracle/_ref/`), and the outputs are committed as fixtures under `tests/golden/` together
with the script that made them (`tests/golden/make_golden.py`).

Modules
  ac_oracle.c / ac.py   range coder (integer, bit-exact)            <- torchac/torchac_backend/torchac.cpp
  cdf.py                mixture CDF -> uint16 table, uniform table   <- torchac/torchac.py:174-213, bitcoding.py:297-323
  dmll.py               parameter extraction, NLL / bpsp             <- criterion/logistic_mixture.py
  net.py                encoder / decoder / prob-classifier forward  <- modules/{multiscale_network,net,edsr,head,prob_clf,quantizer}.py
  bitcoding.py          .l3c container encode / decode               <- bitcoding/bitcoding.py
"""
