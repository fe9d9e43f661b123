"""The register-resident BiWFA kernel (trgt_amd/csrc/wfa_lean.hip) against the oracle: the configurations of the two callers it serves --
utils::align (src/utils/align.rs:14-28: BiWFA, gap-affine 2,5,1, default heuristic, CIGAR) and get_dist
(src/trgt/genotype/genotype_cluster.rs:236-248: score-only BiWFA, edit distance, default heuristic) -- asked for WITHOUT expanded
operations, as trgt_locus_batch does (with them the generic kernel takes the batch).  The lean kernel runs in front of the generic
one and hands over what it does not take, so every comparison below holds whichever kernel finished a job; TRGT_WFA_NO_LEAN=1 contexts
must give the same bytes."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
pytestmark = pytest.mark.gpu


def _modes():
    import wfa_fuzz as F
    from trgt_amd import wfaligner as W
    from oracle import binding as oracle
    A, S, H = W.AlignmentScope, W.MemoryModel, W.Heuristic
    out = []
    for ml in (100, 0):
        for heur in ("default", "none"):
            b = W.WFAligner.builder(A.Alignment, S.MemoryUltraLow).affine(2, 5, 1)
            al = b.build() if heur == "default" else b.with_heuristic(H.none()).build()
            op = oracle.wfa_params(metric="affine", x=2, o1=5, e1=1, memory="ultralow", heuristic=heur, min_length=ml)
            out.append(("affine %s ml%d" % (heur, ml), al, op, ml, ("consensus", "str", "alleles", "generic")))
    for heur in ("default", "none"):
        b = W.WFAligner.builder(A.Score, S.MemoryUltraLow).edit()
        al = b.build() if heur == "default" else b.with_heuristic(H.none()).build()
        out.append(("edit score %s" % heur, al, oracle.wfa_params(metric="edit", scope="score", memory="ultralow", heuristic=heur), None,
                    ("short", "alleles", "consensus", "str")))
        b = W.WFAligner.builder(A.Alignment, S.MemoryUltraLow).edit()
        al = b.build() if heur == "default" else b.with_heuristic(H.none()).build()
        out.append(("edit alignment %s" % heur, al, oracle.wfa_params(metric="edit", memory="ultralow", heuristic=heur), None, ("consensus", "short")))
    return F, out


def test_lean_kernel_matches_oracle():
    F, modes = _modes()
    k = 0
    for name, al, op, ml, kinds in modes:
        for kind in kinds:
            k += 1
            pats, txts = F.gen_pairs(np.random.default_rng(4100 + k), 1500, kind)
            bad = F.run_mode("%s / %s" % (name, kind), al, "end2end", (0, 0, 0, 0), op, pats, txts, min(os.cpu_count() or 1, 32), min_length=ml, want_ops=False)
            assert bad == 0, (name, kind)


def test_lean_and_generic_kernel_agree_byte_for_byte():
    """The same batches through a context with TRGT_WFA_NO_LEAN=1 (generic kernel only)."""
    import wfa_fuzz as F
    from trgt_amd import _lib
    from trgt_amd import wfaligner as W
    A, S = W.AlignmentScope, W.MemoryModel
    os.environ["TRGT_WFA_NO_LEAN"] = "1"
    try:
        ctx_generic = _lib.Context()
    finally:
        del os.environ["TRGT_WFA_NO_LEAN"]
    ctx_mid = _lib.context_with_env(TRGT_WFA_LEAN_MID_TIER=1)  # (round 5: the optional 128-diagonal tier between the 64- and the 256-diagonal kernels)
    for kind, seed in (("consensus", 1), ("alleles", 2), ("str", 3)):
        pats, txts = F.gen_pairs(np.random.default_rng(900 + seed), 1200, kind)
        for build in (lambda c: W.WFAligner.builder(A.Alignment, S.MemoryUltraLow).affine(2, 5, 1).build(c), lambda c: W.WFAligner.builder(A.Score, S.MemoryUltraLow).edit().build(c)):
            a = build(None)
            ra = a.align_end_to_end_batch(pats, txts, want_ops=False)
            for other in (ctx_generic, ctx_mid):
                rb = build(other).align_end_to_end_batch(pats, txts, want_ops=False)
                for f in ("status", "score", "n_match", "span4", "cigar_len"):
                    assert np.array_equal(ra[f], rb[f]), (kind, f)
                for j in range(len(pats)):  # (the slots of the public ABI are worst-case sized: only the first cigar_len entries of each are written)
                    o, n = int(ra["cigar_off"][j]), int(ra["cigar_len"][j])
                    assert np.array_equal(ra["cigar"][o:o + n], rb["cigar"][o:o + n]), (kind, j)
