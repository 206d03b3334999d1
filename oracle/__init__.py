"""TEST INFRASTRUCTURE ONLY (parity unpinned - see oracle/README.md).

CPU restatement of the reference's hot path.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this package; the product (the HIP library and the CLIs built from
speaker-embedding-with-phonetic-information_amd/csrc) never does.
"""
