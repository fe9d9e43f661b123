"""Device-side ingestion (trgt_ingest_params.ingest_device) against the host path on the end-to-end data set of bench.py: 1 000-locus chunks of a
synthetic BAM, host workers vs kernels, 1 / 2 / 3 caller threads (the slots of a reader overlap their file reads, uploads, kernels and
downloads).  Usage: python tools/ingest_dev_probe.py [n_loci] [read_len] [chunk]; TRGT_INGEST_TRACE=1 prints the phase times of every call."""
import os
import sys
import tempfile
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from trgt_amd import ingest, synth_bam  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
read_len = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
d = tempfile.mkdtemp(prefix="trgt_ingdev_")
t0 = time.perf_counter()
ds = synth_bam.write_dataset(d, n_loci=n, read_len=read_len)
print("data set: %d loci, BAM %.1f MB, %.0f MB of records, %.1f s" % (n, ds["bam_bytes"] / 1e6, ds["bases"] / 1e6, time.perf_counter() - t0), flush=True)
rd = ingest.Reader(ds["bam"], ds["fasta"])
firsts = list(range(0, n, chunk))


def run(callers, **kw):
    nxt, lock, out = [0], threading.Lock(), [None] * len(firsts)

    def work():
        while True:
            with lock:
                i = nxt[0]
                nxt[0] += 1
            if i >= len(firsts):
                return
            b = rd.batch(ds["bed"], first_locus=firsts[i], max_loci=chunk, keep_native=True, copy=False, read_names=False, **kw)
            out[i] = int(b["n_reads"])
    th = [threading.Thread(target=work) for _ in range(callers)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    return time.perf_counter() - t0, sum(out)


if os.environ.get("PROBE_DEVICE_ONLY"):  # (for a kernel trace: the device path alone, one caller)
    callers = int(os.environ["PROBE_DEVICE_ONLY"])
    for _ in range(3):
        kw = {"inflate_waves_per_cu": int(os.environ["PROBE_WAVES"])} if os.environ.get("PROBE_WAVES") else {}
        dt, nr = run(callers, ingest_device=0, threads=8, **kw)
        print("device path, %d caller(s): %7.0f loci/s (%d reads)" % (callers, n / dt, nr), flush=True)
    sys.exit(0)
run(1, threads=16)
for threads in (16, 32):
    dt, nr = run(1, threads=threads)
    print("host path, %2d workers          : %7.0f loci/s (%d reads)" % (threads, n / dt, nr), flush=True)
run(1, ingest_device=0, threads=8)
for callers in (1, 2, 3, 4):
    for th in (4, 8):
        dt, nr = run(callers, ingest_device=0, threads=th)
        print("device path, %d callers, %d file threads: %7.0f loci/s (%d reads)  %s" % (callers, th, n / dt, nr, rd.device_stats()), flush=True)
