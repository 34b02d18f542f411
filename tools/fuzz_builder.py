"""Mutation fuzz of the native scene builder (fspt_amd/csrc/scene_build.cpp): 1 200 corrupted copies of two golden asset
sets (OBJ / MTL / scene JSON text: insertions of separators and huge / NaN numbers, deletions, duplications, truncation)
must be built or refused with an error - never crash.  Meant to run under tools/sanitize_cpu.sh (ASan + UBSan)."""
import sys, json, random
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for d in ('', 'tests', 'oracle'):
    sys.path.insert(0, os.path.join(ROOT, d))
import numpy as np
from test_goldens import load_js, stand_in_images
from fspt_amd import scene as S, _lib as L
rng = random.Random(7)
ok = err = 0
for name in ("mtl", "small"):
    z, scene, texts, files = load_js(name)
    imgs = stand_in_images(z)
    keys = sorted(texts) + sorted(files)
    for it in range(600):
        t2, f2 = dict(texts), dict(files)
        k = rng.choice(keys)
        src = t2 if k in t2 else f2
        s = src[k]
        for _ in range(rng.randint(1, 6)):
            op = rng.randint(0, 4)
            if not s: break
            i = rng.randrange(len(s))
            if op == 0: s = s[:i] + rng.choice(["", " ", "\n", "/", "-", "1e309", "nan", "99999999999", "f ", "v ", "usemtl ", "\t", "0", "#"]) + s[i:]
            elif op == 1: j = min(len(s), i + rng.randint(1, 40)); s = s[:i] + s[j:]
            elif op == 2: s = s[:i] + chr(rng.randint(32, 126)) + s[i + 1:]
            elif op == 3: j = min(len(s), i + rng.randint(1, 200)); s = s[:i] + s[i:j] * 2 + s[j:]
            else: s = s[:i]
        src[k] = s
        try:
            S.build_scene_json(scene, t2, f2, imgs, env=z["env"], env_w=int(z["env_w"]), env_h=int(z["env_h"]), focus_rays=z["focus_rays"].tolist())
            ok += 1
        except (L.FsptError, ValueError, KeyError, IndexError, TypeError, AssertionError, OverflowError) as e:
            err += 1
print("mutated inputs:", ok, "built,", err, "refused")
