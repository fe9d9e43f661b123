"""Host-level driver: N contexts in one process, fed from a dynamic queue of locus chunks.

The reference runs one rayon task per locus and hands results to a writer thread over a channel (src/commands/genotype.rs:140-187);
a host that binds libtrgt_hip does the same one level up: one worker thread per context (a context is single-threaded and owns its
streams, include/trgt_hip.h), every worker pulling the next chunk of ~4k loci from a shared queue -- dynamic, so that a chunk of
long alleles (cfg3 / cfg5) does not hold the others up -- and writing its results into that chunk's own output slot.  Contexts may
sit on different GPUs (one or more per device) or share one: two contexts on ONE device overlap the host-bound tail of a call
(consensus repair, HMM collection: ~3 of 8 ms on the cfg2 workload) with the flank location of the next chunk.

Loci are independent, so there is no exchange between workers, and results are byte-identical to the single-context ones whatever
the number of contexts or the order in which chunks complete (tests/test_driver_gpu.py, bench.py --contexts).
"""
import queue
import threading

from . import _lib, locus


class ChunkDriver:
    """devices: one entry per context (ordinals may repeat: several contexts on one GPU)."""

    def __init__(self, devices=(0,), params=None, context_factory=None, run_fn=None):
        self.params = params or locus.Params()
        self._make = context_factory or (lambda dev: _lib.Context(dev))
        self._run = run_fn or (lambda ctx, chunk, params, kw: locus.run_batch(chunk, params, ctx, **kw))
        self.devices = list(devices)
        self.contexts = [self._make(d) for d in self.devices]
        self.chunks_by_context = [0] * len(self.contexts)

    def close(self):
        for c in self.contexts:
            if hasattr(c, "close"):
                c.close()
        self.contexts = []

    def run(self, chunks, per_chunk_kwargs=None, device_kwargs=None, worker_kwargs=None):
        """chunks: list of packed batches (dicts in the trgt_locus_batch_in layout).  per_chunk_kwargs[i]: extra arguments of
        locus.run_batch for chunk i (outputs=..., and flank_dev= / reads_dev= when every context is on the same device);
        device_kwargs(i, device): the same, computed by the worker for the device its context sits on (multi-GPU: each device needs
        its own HBM copy of a chunk's bytes); worker_kwargs(w): the same per worker (e.g. a reusable BatchOutputs of its own).
        Returns the outputs in chunk order.  Exceptions of a worker are re-raised here."""
        n = len(chunks)
        kws = per_chunk_kwargs or [{} for _ in range(n)]
        results = [None] * n
        errors = []
        q = queue.Queue()
        for i in range(n):
            q.put(i)

        def worker(w):
            ctx = self.contexts[w]
            while not errors:
                try:
                    i = q.get_nowait()
                except queue.Empty:
                    return
                try:
                    kw = dict(kws[i])
                    if device_kwargs is not None:
                        kw.update(device_kwargs(i, self.devices[w]))
                    if worker_kwargs is not None:
                        kw.update(worker_kwargs(w))
                    results[i] = self._run(ctx, chunks[i], self.params, kw)
                    self.chunks_by_context[w] += 1
                except Exception as e:  # noqa: BLE001 -- handed to the caller
                    errors.append(e)
                    return

        threads = [threading.Thread(target=worker, args=(w,), daemon=True) for w in range(len(self.contexts))]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return results


def split_batch(batch, chunk_loci):
    """Cut a packed batch into chunks of at most chunk_loci loci (views / small copies of the offset tables; the blobs are shared:
    a chunk's offsets keep pointing into the parent's blobs, so device copies of the parent's blobs serve every chunk)."""
    import numpy as np
    nl = int(batch["n_loci"])
    out = []
    for a in range(0, nl, chunk_loci):
        b = min(nl, a + chunk_loci)
        r0, r1 = int(batch["locus_read_begin"][a]), int(batch["locus_read_begin"][b])
        m0, m1 = int(batch["set_motif_begin"][a]), int(batch["set_motif_begin"][b])
        c = dict(n_loci=b - a, n_reads=r1 - r0, flank_blob=batch["flank_blob"], tr_blob=batch["tr_blob"], motif_blob=batch["motif_blob"],
                 read_blob=batch["read_blob"])
        for k in ("lf_off", "lf_len", "rf_off", "rf_len", "tr_off", "tr_len", "ploidy"):
            c[k] = np.ascontiguousarray(batch[k][a:b])
        c["locus_read_begin"] = np.ascontiguousarray(batch["locus_read_begin"][a:b + 1] - np.uint64(r0))
        c["read_off"] = np.ascontiguousarray(batch["read_off"][r0:r1])
        c["read_len"] = np.ascontiguousarray(batch["read_len"][r0:r1])
        c["set_motif_begin"] = np.ascontiguousarray(batch["set_motif_begin"][a:b + 1] - np.uint32(m0))
        c["motif_off"] = np.ascontiguousarray(batch["motif_off"][m0:m1 + 1])
        if batch.get("genotyper") is not None:
            c["genotyper"] = np.ascontiguousarray(batch["genotyper"][a:b])
        if batch.get("read_qual") is not None and len(batch["read_qual"]):
            c["read_qual"] = np.ascontiguousarray(batch["read_qual"][r0:r1])
        # the per-read fields genotype_flank reads (tr.rs:69-75): without them a chunk would skip the flank re-genotyping
        for k in ("hp_tag", "start_offset", "end_offset"):
            if batch.get(k) is not None:
                c[k] = np.ascontiguousarray(batch[k][r0:r1])
        if batch.get("mismatch_off") is not None and batch.get("mismatch_offsets") is not None:
            mo = batch["mismatch_off"]
            m0o, m1o = int(mo[r0]), int(mo[r1])
            c["mismatch_off"] = np.ascontiguousarray(mo[r0:r1 + 1] - np.uint64(m0o))
            c["mismatch_offsets"] = np.ascontiguousarray(np.concatenate([batch["mismatch_offsets"][m0o:m1o], np.zeros(1, np.int32)]).astype(np.int32))
        # (4-bit reads: the chunk's read_off still points into the parent's PACKED blob)
        if batch.get("read_encoding"):
            c["read_encoding"] = int(batch["read_encoding"])
        out.append(c)
    return out
