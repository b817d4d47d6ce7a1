"""ORACLE package — test infrastructure, NOT product code.

CPU restatements (plain PyTorch fp32 / pure Python ints) of the reference algorithms on MMGT's Stage-2 denoising
path.  Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  `mmgt_amd` never imports it.
"""
