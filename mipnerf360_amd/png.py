"""Dependency-free PNG writer (the reference writes frames with cv2, which is not part of this stack):
uint8 [H,W,3] (RGB) or [H,W] (grey) -> PNG bytes, zlib-compressed, no filtering."""
from __future__ import annotations

import struct
import zlib

import numpy as np


def encode_png(img: np.ndarray) -> bytes:
    img = np.ascontiguousarray(img)
    if img.dtype != np.uint8 or img.ndim not in (2, 3) or (img.ndim == 3 and img.shape[2] not in (3, 4)):
        raise ValueError("encode_png expects uint8 [H,W], [H,W,3] or [H,W,4]")
    h, w = img.shape[:2]
    color = {2: 0, 3: 2, 4: 6}[2 if img.ndim == 2 else img.shape[2]]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, -1)], axis=1).tobytes()  # filter byte 0 per row

    def chunk(tag: bytes, data: bytes) -> bytes:
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def write_png(path: str, img: np.ndarray) -> None:
    with open(path, "wb") as f:
        f.write(encode_png(img))
