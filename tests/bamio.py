"""Minimal writers for unaligned BAM (BGZF) and SAM text -- test infrastructure for the BAM/SAM input
path of the command line (fixtures in tests/golden/ are made with these and run through the reference)."""
from __future__ import annotations

import struct
import zlib

NT16 = "=ACMGRSVTWYHKDBN"
_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _bgzf_block(data: bytes) -> bytes:
    co = zlib.compressobj(6, zlib.DEFLATED, -15)
    comp = co.compress(data) + co.flush()
    bsize = len(comp) + 25                      # total block size - 1
    head = b"\x1f\x8b\x08\x04" + b"\0\0\0\0" + b"\0\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
    return head + comp + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def bgzf(data: bytes, block: int = 0xFF00) -> bytes:
    out = bytearray()
    for i in range(0, len(data), block):
        out += _bgzf_block(data[i:i + block])
    return bytes(out) + _EOF


def bam_record(name: bytes, seq: bytes, qual: bytes | None, flag: int = 4, aux: bytes = b"") -> bytes:
    """qual: Phred+33 text of len(seq), or None for 'absent' (0xFF fill)."""
    n = len(seq)
    code = {c: i for i, c in enumerate(NT16)}
    packed = bytearray((n + 1) // 2)
    for i, ch in enumerate(seq.decode().upper()):
        v = code.get(ch, 15)
        packed[i >> 1] |= v << 4 if (i & 1) == 0 else v
    q = bytes([0xFF]) * n if qual is None else bytes(c - 33 for c in qual)
    rn = name + b"\0"
    core = struct.pack("<iiBBHHHIiii", -1, -1, len(rn), 255, 4680, 0, flag, n, -1, -1, 0)
    body = core + rn + bytes(packed) + q + aux
    return struct.pack("<i", len(body)) + body


def write_bam(path: str, reads, header_text: bytes = b"@HD\tVN:1.6\tSO:unknown\n@RG\tID:x\tPL:PACBIO\n", aux_every: int = 3,
              block: int = 0xFF00):
    body = bytearray(b"BAM\1" + struct.pack("<i", len(header_text)) + header_text + struct.pack("<i", 0))
    for i, (name, seq, qual) in enumerate(reads):
        aux = b"npi" + struct.pack("<i", 7) + b"RGZx\0" if aux_every and i % aux_every == 0 else b""
        body += bam_record(name.split()[0], seq, qual, aux=aux)
    open(path, "wb").write(bgzf(bytes(body), block))


def write_sam(path: str, reads, header_text: bytes = b"@HD\tVN:1.6\tSO:unknown\n"):
    with open(path, "wb") as f:
        f.write(header_text)
        for i, (name, seq, qual) in enumerate(reads):
            tags = b"\tnp:i:7" if i % 3 == 0 else b""
            f.write(b"\t".join([name.split()[0], b"4", b"*", b"0", b"255", b"*", b"*", b"0", b"0", seq,
                                qual if qual is not None else b"*"]) + tags + b"\n")
