"""Device-side BGZF inflate (trgt_inflate_blocks / trgt_ingest_params.inflate_device) on a synthetic BAM: ingestion of 1000-locus chunks
with and without it, phase times (TRGT_INGEST_TRACE).  tools/inflate_probe.py [n_loci] [read_len]"""
import os, sys, time, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TRGT_INGEST_TRACE"] = "1"
from trgt_amd import ingest, synth_bam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rl = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
d = tempfile.mkdtemp(prefix="trgt_infl_")
try:
    ds = synth_bam.write_dataset(d, n_loci=n, read_len=rl)
    rd = ingest.Reader(ds["bam"], ds["fasta"])
    for dev in (-1, 0, 0, -1, 0):
        t0 = time.perf_counter()
        for a in range(0, n, 1000):
            rd.batch(ds["bed"], first_locus=a, max_loci=1000, keep_native=True, copy=False, read_names=False, threads=32, keep_bam4=1, inflate_device=dev)
        print("inflate_device %2d: %.0f loci/s" % (dev, n / (time.perf_counter() - t0)), flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
