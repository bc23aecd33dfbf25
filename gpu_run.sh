./tools/ac_microbench
