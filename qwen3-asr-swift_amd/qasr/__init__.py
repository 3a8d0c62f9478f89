"""qasr -- host-side mirror of the reference's Qwen3ASR API over libqasr.so (HIP, gfx950).

The compute path lives entirely in the C-ABI library (csrc/); this package is the thin
Python harness used by tests and bench.py.  There is no CPU fallback: importing `qasr.model`
on a machine without the built library or without a GPU raises.
"""
