"""`.fspt` scene blob: the arrays main.js uploads (SceneArrays), in one little-endian file, so a scene built
once (e.g. by the reference's JS pipeline or the native builder) can be rendered by any host.

    header  : magic b"FSPT" | u32 version (1) | u32 n_sections
    section : 8-byte ASCII name (NUL padded) | u32 dtype (0 f32, 1 u8, 2 u32) | u32 reserved | u64 n_elements | data
              (data padded to 16 bytes)
    sections: bvh tri mat norm uv atlas env bins (env optional) + meta (u32: atlas_res atlas_layers env_w env_h
              leaf_size depth)
The same layout is read/written by fspt_amd/js/fspt.js (saveBlob / loadBlob).
"""
import struct

import numpy as np

from .scene import SceneArrays

MAGIC = b"FSPT"
VERSION = 1
_DT = {0: np.float32, 1: np.uint8, 2: np.uint32}
_CODE = {np.dtype(np.float32): 0, np.dtype(np.uint8): 1, np.dtype(np.uint32): 2}


def save(path, a):
    secs = [("bvh", a.bvh), ("tri", a.tri), ("mat", a.mat), ("norm", a.norm), ("uv", a.uv), ("atlas", a.atlas),
            ("bins", a.bins.astype(np.uint32)),
            ("meta", np.array([a.atlas_res, a.atlas_layers, a.env_w if a.env is not None else 0,
                               a.env_h if a.env is not None else 0, a.leaf_size, a.depth], np.uint32))]
    if a.env is not None:
        secs.append(("env", a.env))
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<II", VERSION, len(secs)))
        for name, arr in secs:
            arr = np.ascontiguousarray(arr).reshape(-1)
            f.write(name.encode().ljust(8, b"\0") + struct.pack("<IIQ", _CODE[arr.dtype], 0, arr.size))
            raw = arr.tobytes()
            f.write(raw + b"\0" * (-len(raw) % 16))


def load(path):
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:4] != MAGIC:
        raise ValueError("not an .fspt blob")
    version, n = struct.unpack_from("<II", buf, 4)
    if version != VERSION:
        raise ValueError(f"unsupported .fspt version {version}")
    off, secs = 12, {}
    for _ in range(n):
        name = buf[off:off + 8].rstrip(b"\0").decode()
        code, _, count = struct.unpack_from("<IIQ", buf, off + 8)
        off += 24
        dt = np.dtype(_DT[code])
        nbytes = count * dt.itemsize
        if off + nbytes > len(buf):
            raise ValueError("truncated .fspt blob")
        secs[name] = np.frombuffer(buf, dtype=dt, count=count, offset=off).copy()
        off += nbytes + (-nbytes % 16)
    m = secs["meta"]
    env = secs.get("env")
    return SceneArrays(bvh=secs["bvh"], tri=secs["tri"], mat=secs["mat"], norm=secs["norm"], uv=secs["uv"],
                       atlas=secs["atlas"], atlas_res=int(m[0]), atlas_layers=int(m[1]), env=env, env_w=int(m[2]),
                       env_h=int(m[3]), bins=secs["bins"], leaf_size=int(m[4]), depth=int(m[5]))
