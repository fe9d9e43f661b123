"""trgt_amd -- MI355X (gfx950) implementation of TRGT's per-locus alignment / DP hot path.

Host-side mirror of the reference's interface for this path (PacificBiosciences/trgt v3.0.0):
  trgt_amd.wfaligner  WFAligner / WFAlignerBuilder    (src/wfaligner.rs)
  trgt_amd.hmm        build_hmm / Hmm / label_with_hmm (src/hmm/, src/trgt/workflows/tr.rs:454-492)
  trgt_amd.locus      Locus / Params / analyze_tr      (src/trgt/workflows/tr.rs:24-109)
All compute goes through the C ABI of libtrgt_hip.so (include/trgt_hip.h); there is no CPU fallback.
"""
from ._lib import TrgtHipError, build_extension, context, lib  # noqa: F401
