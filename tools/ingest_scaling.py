#!/usr/bin/env python3
"""Ingestion rate of the native reader against threads and chunk size on a synthetic data set (trgt_amd/synth_bam.py).
  python tools/ingest_scaling.py [n_loci] [read_len]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tempfile

from trgt_amd import ingest, synth_bam

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
d = tempfile.mkdtemp(prefix="trgt_ing_")
ds = synth_bam.write_dataset(d, n_loci=n, read_len=rl)
rd = ingest.Reader(ds["bam"], ds["fasta"])
rd.batch(ds["bed"], threads=16, keep_native=True, copy=False, read_names=False)
for chunk in (1000, n):
    for th in (1, 8, 16, 32, 64, 128, 256):
        if th > (os.cpu_count() or 8):
            continue
        t0 = time.perf_counter()
        for a in range(0, n, chunk):
            rd.batch(ds["bed"], first_locus=a, max_loci=chunk, threads=th, keep_native=True, copy=False, read_names=False, keep_bam4=1)
        dt = time.perf_counter() - t0
        print("chunk %5d threads %3d: %8.0f loci/s  (%.1f MB/s of records)" % (chunk, th, n / dt, ds["bases"] / 1e6 / dt), flush=True)
