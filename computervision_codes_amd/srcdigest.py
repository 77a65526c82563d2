"""sha256 (first 16 hex digits) over the HIP sources a measured kernel set is built from.  `profiles/traffic.json` / `mfma_util.json` record it
when the PMC passes are collected (tools/collect_*.py); `bench.py` publishes a committed counter figure only while it still matches the sources
of the library it runs, and says which collection it is.  No torch import: the collectors use it too."""
import hashlib
import os

CONV_SOURCES = ("igemm_conv.hip", "bottleneck_fused.hip", "chain_gemm.hip", "mt4_common.h")            # every conv launch of the spatial extractor
TCN_SOURCES = ("tcn_kernels.hip", "igemm_conv.hip", "mt4_common.h")                  # the launches of a Temporal_tenco forward (fpn_topdown lives in tcn_kernels.hip)


def kernels_digest(files=CONV_SOURCES) -> str:
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for f in sorted(files):
        h.update(f.encode())
        h.update(open(os.path.join(here, f), "rb").read())
    return h.hexdigest()[:16]


def library_sources():
    """every file `libmt4hip.so` is built from: csrc/*.hip, csrc/*.h and the C-ABI header (names relative to csrc/)"""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    return tuple(sorted(f for f in os.listdir(here) if f.endswith((".hip", ".h")))) + ("../../include/mt4hip.h",)


def library_digest() -> str:
    """digest of ALL library sources.  The Makefile bakes it into `libmt4hip.so` (`mt4_source_digest()`); `_lib.py` refuses a library whose
    baked digest is not the digest of the sources beside it (a stale `.so` / `.o` would silently detach every committed PMC figure and every
    test from the code in the tree) unless MT4_ALLOW_STALE=1."""
    return kernels_digest(library_sources())


if __name__ == "__main__":
    print(library_digest())
