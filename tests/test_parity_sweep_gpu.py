"""Every locus of a few thousand-locus synthetic catalogs (configs 2, 4 and 5) through trgt_locus_batch and through the oracle,
compared as whole text records (spans of every read, allele sequences, kept reads + classification, ALLR, SD, MC, MS, AP).  The same
comparison over 10^6 + 2*10^5 + 6*10^4 loci is tests/tools/parity_sweep.py (result of the last run: profiles/r01_parity_sweep.txt)."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu


def _tool():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "parity_sweep.py")
    spec = importlib.util.spec_from_file_location("parity_sweep", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("config,n,first", [(2, 3000, 123456), (4, 2000, 777000), (5, 1000, 31000)])
def test_catalog_slice_matches_oracle_record_for_record(config, n, first):
    import torch
    from trgt_amd import locus, synth
    t = _tool()
    b = synth.generate(n, first_locus=first, config=config)
    out = locus.run_batch(b, flank_dev=torch.from_numpy(b["flank_blob"]).cuda(), reads_dev=torch.from_numpy(b["read_blob"]).cuda())
    got = t.gpu_records(b, out)
    ref = t.oracle.locus_records(b, 0, n, min(os.cpu_count() or 1, 64))
    assert len(got) == len(ref) == n
    assert all(len(r) > 40 for r in ref)  # real records, not empty strings
    bad = [l for l in range(n) if got[l] != ref[l]]
    assert not bad, (first + bad[0], got[bad[0]][:400], ref[bad[0]][:400])


def test_shortcut_fuzz_slice_matches_oracle():
    """A round of tests/tools/shortcut_fuzz.py (low-complexity flanks with 1-3 substitutions, one-base gaps at the margins and inside
    runs, decoy copies): the two shortcuts of the seed search must settle a share of the fallback alignments and change no record."""
    import sys
    import numpy as np
    import torch
    from trgt_amd import locus
    tools = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools")
    sys.path.insert(0, tools)
    try:
        import shortcut_fuzz as sf
    finally:
        sys.path.remove(tools)
    rng = np.random.default_rng(20250929)
    n = 1500
    b = locus.pack([sf.make_locus(rng) for _ in range(n)])
    out = locus.run_batch(b, flank_dev=torch.from_numpy(b["flank_blob"]).cuda(), reads_dev=torch.from_numpy(b["read_blob"]).cuda())
    got = sf.gpu_records(b, out)
    ref = sf.oracle.locus_records(b, 0, n, min(os.cpu_count() or 1, 64))
    bad = [l for l in range(n) if got[l] != ref[l]]
    assert not bad, (bad[0], got[bad[0]][:400], ref[bad[0]][:400])
    assert int(out.stats[21]) > int(out.stats[0]) // 10  # (a sixth of these fallback alignments take a shortcut)
