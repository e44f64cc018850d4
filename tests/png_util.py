"""A minimal PNG ENCODER for the tests (the product only decodes): any colour type / bit depth the decoder
supports, a chosen scanline filter per row, IDAT split into several chunks, optional ancillary chunks."""
import struct
import zlib

import numpy as np


def _chunk(typ, body):
    return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body))


def _filter_row(ft, cur, up, bpp):
    cur = cur.astype(np.int64)
    up = up.astype(np.int64)
    a = np.concatenate([np.zeros(bpp, np.int64), cur[:-bpp]]) if bpp < cur.size else np.zeros_like(cur)
    c = np.concatenate([np.zeros(bpp, np.int64), up[:-bpp]]) if bpp < cur.size else np.zeros_like(cur)
    if ft == 0:
        pred = np.zeros_like(cur)
    elif ft == 1:
        pred = a
    elif ft == 2:
        pred = up
    elif ft == 3:
        pred = (a + up) // 2
    else:
        p = a + up - c
        pa, pb, pc = np.abs(p - a), np.abs(p - up), np.abs(p - c)
        pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
    return ((cur - pred) & 255).astype(np.uint8)


def encode_png(samples, color_type, bit_depth, filters=(0, 1, 2, 3, 4), palette=None, idat_split=3, extra_chunks=(), interlace=0):
    """samples: (rows, cols, channels) integer array of sample values (palette indices for colour type 3)."""
    samples = np.asarray(samples)
    if samples.ndim == 2:
        samples = samples[..., None]
    h, w, ch = samples.shape
    assert ch == {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    if bit_depth == 16:
        s = samples.astype(">u2").reshape(h, w * ch)
        rows = s.view(np.uint8).reshape(h, w * ch * 2)
    elif bit_depth == 8:
        rows = samples.astype(np.uint8).reshape(h, w * ch)
    else:
        per = 8 // bit_depth
        pad = (-w) % per
        v = np.concatenate([samples[..., 0].astype(np.uint8), np.zeros((h, pad), np.uint8)], axis=1)
        bits = ((v[..., None] >> np.arange(bit_depth - 1, -1, -1)) & 1).astype(np.uint8).reshape(h, -1)
        rows = np.packbits(bits, axis=1)
    bpp = max(1, ch * bit_depth // 8)
    raw = bytearray()
    prev = np.zeros(rows.shape[1], np.uint8)
    for y in range(h):
        ft = filters[y % len(filters)]
        raw.append(ft)
        raw += _filter_row(ft, rows[y], prev, bpp).tobytes()
        prev = rows[y]
    comp = zlib.compress(bytes(raw), 6)
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bit_depth, color_type, 0, 0, interlace))
    for typ, body in extra_chunks:
        out += _chunk(typ, body)
    if palette is not None:
        out += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    n = max(1, len(comp) // idat_split)
    for i in range(0, len(comp), n):
        out += _chunk(b"IDAT", comp[i:i + n])
    return out + _chunk(b"IEND", b"")
