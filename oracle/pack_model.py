"""numpy restatement of the packed on-disk format (test infrastructure only).

reference src/packed_sa.rs: header (magic "SA4x" LE :6-7, length, bincode Vec<u8> length prefix
:90-124), bits = ceil(log2(length)) (:127-129), blocks of 128 integers packed with
`bitpacking::BitPacker4x`, last partial block zero padded then right-trimmed of zero bytes (:36-46).

PARITY UNPINNED AT BYTE LEVEL: the block layout is the external crate `bitpacking 0.8`'s (not under
/root/reference).  It is restated here from its published description (SIMD-BP128 vertical layout:
integer 4 i + c is row i of lane c; each lane is a little-endian bit stream, row i at bit i * bits;
output register j holds 32-bit word j of lanes 0..3).  The reference's own test pins only the round
trip (src/tests.rs:61-76).
"""
import struct

import numpy as np

MAGIC = 2016690515          # src/packed_sa.rs:7


def sa_bits(length: int) -> int:                 # src/packed_sa.rs:127-129
    return (max(length, 1) - 1).bit_length()


def pack_block(vals128, bits):
    out = np.zeros(bits * 4, dtype=np.uint32)
    for c in range(4):
        stream = 0
        for i in range(32):
            stream |= int(vals128[4 * i + c]) << (i * bits)
        for j in range(bits):
            out[4 * j + c] = (stream >> (32 * j)) & 0xFFFFFFFF
    return out


def pack(sa) -> bytes:
    sa = np.asarray(sa, dtype=np.uint32)
    length = sa.size
    bits = sa_bits(length)
    data = bytearray()
    if bits:
        full = length // 128
        for b in range(full):
            data += pack_block(sa[128 * b:128 * b + 128], bits).astype("<u4").tobytes()
        if length % 128:
            chunk = np.zeros(128, dtype=np.uint32)
            chunk[:length % 128] = sa[128 * full:]
            buf = pack_block(chunk, bits).astype("<u4").tobytes()
            data += buf.rstrip(b"\x00")
    return struct.pack("<IIQ", MAGIC, length, len(data)) + bytes(data)


def unpack(blob: bytes) -> np.ndarray:
    magic, length, dl = struct.unpack_from("<IIQ", blob, 0)
    assert magic == MAGIC and dl == len(blob) - 16
    bits = sa_bits(length)
    out = np.zeros(length, dtype=np.uint32)
    if bits == 0:
        return out
    blocks = (length + 127) // 128
    data = blob[16:] + b"\x00" * (blocks * bits * 16 - dl)
    words = np.frombuffer(data, dtype="<u4")
    for b in range(blocks):
        w = words[b * bits * 4:(b + 1) * bits * 4]
        for c in range(4):
            stream = 0
            for j in range(bits):
                stream |= int(w[4 * j + c]) << (32 * j)
            for i in range(32):
                idx = 128 * b + 4 * i + c
                if idx < length:
                    out[idx] = (stream >> (i * bits)) & ((1 << bits) - 1)
    return out
