"""Native writers (trgt_amd/csrc/writers.hip through the C ABI): the VCF of VcfWriter (src/trgt/writers/write_vcf.rs:19-397, incl. the AM
field of get_meth, tr.rs:196-262, 363-398) and the spanning-reads BAM of BamWriter (src/trgt/writers/write_bam.rs:33-144) for batches that
came through the native ingestion (trgt_amd/ingest.py) and trgt_locus_batch.  tests/pyvcf.py is the Python mirror of one VCF record."""
import ctypes as C

import numpy as np

from . import _lib
from .ingest import IngestBatch


class WriterParams(C.Structure):
    _fields_ = [("output_flank_len", C.c_int32), ("sample_name", C.c_char_p), ("program", C.c_char_p), ("version", C.c_char_p),
                ("command_line", C.c_char_p), ("keep_unmapped_flag", C.c_int32), ("threads", C.c_int32), ("bam_compress_level", C.c_int32),
                ("deflate_device", C.c_int32), ("write_behind", C.c_int32)]


class Writer:
    def __init__(self, reader, vcf_path, bam_path=None, output_flank_len=50, sample_name="sample", program="trgt", version="3.0.0",
                 command_line="", keep_unmapped_flag=1, threads=0, bam_compress_level=6, deflate_device=-1, write_behind=0):
        L = _lib.lib()
        L.trgt_writer_open.argtypes = [C.c_void_p, C.POINTER(WriterParams), C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p)]
        L.trgt_writer_write.argtypes = [C.c_void_p, C.POINTER(IngestBatch), C.c_void_p]
        L.trgt_writer_close.argtypes = [C.c_void_p]
        L.trgt_writer_last_error.argtypes = [C.c_void_p]
        L.trgt_writer_last_error.restype = C.c_char_p
        self._L = L
        self._keep = [s.encode() for s in (sample_name, program, version, command_line)]
        p = WriterParams(output_flank_len, *self._keep, int(keep_unmapped_flag), int(threads), int(bam_compress_level), int(deflate_device), int(write_behind))
        self.handle = C.c_void_p()
        rc = L.trgt_writer_open(reader.handle, C.byref(p), str(vcf_path).encode(), str(bam_path).encode() if bam_path else None, C.byref(self.handle))
        if rc != 0:
            msg = L.trgt_writer_last_error(self.handle).decode() if self.handle else "trgt_writer_open failed"
            if self.handle:
                L.trgt_writer_close(self.handle)
                self.handle = C.c_void_p()
            raise _lib.TrgtHipError("trgt_writer_open: %s" % msg)

    def write(self, batch, outputs):
        """batch: a dict of ingest.Reader.batch(..., keep_native=True); outputs: the locus.BatchOutputs trgt_locus_batch filled for it"""
        rc = self._L.trgt_writer_write(self.handle, batch["_native"].handle, C.addressof(outputs.c_out))
        if rc != 0:
            raise _lib.TrgtHipError("trgt_writer_write: %s" % self._L.trgt_writer_last_error(self.handle).decode())

    def device_stats(self):
        """trgt_writer_device_stats: BGZF blocks of the spanning BAM deflated on the device / declined by it / deflated by zlib for lack of a
        device or of blocks (a flush of fewer than 16)."""
        v = (C.c_int64 * 3)()
        self._L.trgt_writer_device_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        self._L.trgt_writer_device_stats.restype = None
        self._L.trgt_writer_device_stats(self.handle, v)
        return dict(device=int(v[0]), declined=int(v[1]), host=int(v[2]))

    def close(self):
        if self.handle:
            rc = self._L.trgt_writer_close(self.handle)
            self.handle = C.c_void_p()
            if rc != 0:
                raise _lib.TrgtHipError("trgt_writer_close failed")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def deflate_blocks(ctx, datas, cap=0xFF00):
    """trgt_deflate_blocks: blocks of at most 65536 bytes -> (list of raw DEFLATE streams, None for a block the device declined).  The
    device-side stand-in for the zlib deflate inside htslib's bgzf_write; every stream is one final block with fixed Huffman codes."""
    L = _lib.lib()
    n = len(datas)
    src_off = np.zeros(n, np.uint64); src_len = np.zeros(n, np.uint32); dst_off = np.zeros(n, np.uint64)
    dst_cap = np.full(n, cap, np.uint32); dst_len = np.zeros(n, np.uint32)
    so = do = 0
    for i, d in enumerate(datas):
        src_off[i], src_len[i], dst_off[i] = so, len(d), do
        so += (len(d) + 15) & ~15
        do += (cap + 8 + 63) & ~63
    src = np.zeros(so + 16, np.uint8)
    for i, d in enumerate(datas):
        src[int(src_off[i]):int(src_off[i]) + len(d)] = np.frombuffer(d, np.uint8)
    dst = np.full(do + 64, 0xA5, np.uint8)
    L.trgt_deflate_blocks.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 7
    L.trgt_deflate_blocks.restype = C.c_int
    ctx.check(L.trgt_deflate_blocks(ctx.handle, n, src.ctypes.data, src_off.ctypes.data, src_len.ctypes.data, dst.ctypes.data, dst_off.ctypes.data,
                                    dst_cap.ctypes.data, dst_len.ctypes.data))
    out = []
    for i in range(n):
        a, k = int(dst_off[i]), int(dst_len[i])
        out.append(dst[a:a + k].tobytes() if k > 0 else None)
    return out
