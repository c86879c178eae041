"""oracle -- CPU restatement of the reference's BwdTrans arithmetic.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package;
the product path (gpu-benchmarking_amd/) must never do so.
"""
from .oracle import *  # noqa: F401,F403
