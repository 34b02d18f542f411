"""Driver for the SwiftShader GLSL harness (build-container tooling).

Reads the reference's shaders from /root/reference/shader AT RUN TIME (nothing
is copied into the repo), splices the `#define`s exactly as
main.js:873-877 (commitPreprocessor) does, applies the documented
quad-replication substitutions to a copy of tracer.fs main() (SURVEY.md App.
B.4) and exposes drawCamera / drawTracer / readback to Python.
"""
import ctypes as C
import math
import os
import re
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
SWIFTSHADER = "/usr/local/lib/python3.10/dist-packages/kaleido/executable/bin/swiftshader"
_F = C.POINTER(C.c_float)


def available():
    return os.path.isdir(REF) and os.path.exists(os.path.join(SWIFTSHADER, "libGLESv2.so"))


def _build():
    out = os.path.join(HERE, "_build", "libglslharness.so")
    src = os.path.join(HERE, "harness.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", out, src, "-ldl"])
    return out


def pad_buffer(buf, per_element, channels):
    """padBuffer (main.js:143-154): pads with -1 to a width x height rectangle."""
    buf = np.asarray(buf, dtype=np.float32).reshape(-1)
    num_pixels = buf.size / channels
    root = math.sqrt(num_pixels)
    width = int(math.ceil(root / per_element) * per_element)
    height = int(math.ceil(num_pixels / width))
    out = np.full(channels * width * height, -1.0, dtype=np.float32)
    out[:buf.size] = buf
    return out, width, height


def pad_bvh(bvh):
    """padBuffer + maskBVHBuffer (main.js:272-282): the -1 padding of the int words
    becomes int bits 0xFFFFFFFF, the box words stay float -1."""
    n = bvh.size
    out, w, h = pad_buffer(bvh, 3, 3)
    pad = out[n:].reshape(-1)
    iv = out.view(np.int32)
    k = np.arange(n, out.size)
    iv[k[(k % 9) < 3]] = -1
    return out, w, h


def read_shader(name):
    return open(os.path.join(REF, "shader", name)).read()


def tracer_source(n_bins, leaf_size=4, replicate=True, num_bounces=None, main_override=None, post=None):
    """tracer.fs with the preprocessor lines of main.js:293/299,403-405,895 spliced after
    line 1; optionally NUM_BOUNCES changed (tracer.fs:9 is a compile-time constant), the
    quad-replication substitutions, or a replacement main() for instrumented probes."""
    src = read_shader("tracer.fs")
    lines = src.split("\n")
    defs = [f"#define ENV_BINS {n_bins}", "#define NUM_LIGHT_RANGES 1", f"#define LEAF_SIZE {leaf_size}"]
    lines[1:1] = defs
    src = "\n".join(lines)
    if num_bounces is not None:
        src, n = re.subn(r"const int NUM_BOUNCES = \d+;", f"const int NUM_BOUNCES = {num_bounces};", src)
        assert n == 1
    if main_override is not None:
        i = src.index("void main(void) {")
        src = src[:i] + main_override
    if replicate and main_override is None:
        subs = [
            ("vec2 dims = vec2(textureSize(fbTex, 0));\n  seed = randBase + gl_FragCoord.x + gl_FragCoord.y * dims.x;",
             "vec2 FC = floor(gl_FragCoord.xy * 0.5) + vec2(0.5);\n  vec2 dims = vec2(textureSize(fbTex, 0)) * 0.5;\n"
             "  seed = randBase + FC.x + FC.y * dims.x;"),
            ("texelFetch(cameraPosTex, ivec2(gl_FragCoord), 0)", "texelFetch(cameraPosTex, ivec2(FC), 0)"),
            ("texelFetch(cameraDirTex, ivec2(gl_FragCoord), 0)", "texelFetch(cameraDirTex, ivec2(FC), 0)"),
        ]
        for a, b in subs:
            assert src.count(a) == 1, a
            src = src.replace(a, b)
    if post is not None:
        src = post(src)  # instrumentation wrapped AROUND the reference's functions (tools/make_goldens.py path_replay)
    return src


def tracer_test_source(n_bins, leaf_size=4, replicate=True):
    """bvh_test.fs (what main.js:879-883 swaps in for tracer.fs under mode=test) with the same preprocessor
    lines; only the two camera-texture fetches of main() are re-pointed for the quad replication."""
    src = read_shader("bvh_test.fs")
    lines = src.split("\n")
    lines[1:1] = [f"#define ENV_BINS {n_bins}", "#define NUM_LIGHT_RANGES 1", f"#define LEAF_SIZE {leaf_size}"]
    src = "\n".join(lines)
    if replicate:
        a = "void main(void) {"
        assert src.count(a) == 1
        src = src.replace(a, a + "\n  vec2 FC = floor(gl_FragCoord.xy * 0.5) + vec2(0.5);")
        for tex in ("cameraPosTex", "cameraDirTex"):
            old = f"texelFetch({tex}, ivec2(gl_FragCoord), 0)"
            assert src.count(old) == 1, old
            src = src.replace(old, f"texelFetch({tex}, ivec2(FC), 0)")
    return src


class GlslRef:
    def __init__(self):
        self.lib = C.CDLL(_build())
        self.lib.gh_error.restype = C.c_char_p
        self.lib.gh_renderer.restype = C.c_char_p
        self.lib.gh_extensions.restype = C.c_char_p
        self._ck(self.lib.gh_init(SWIFTSHADER.encode()))
        self.renderer = self.lib.gh_renderer().decode()
        self._ck(self.lib.gh_camera_program(read_shader("camera.vs").encode(), read_shader("camera.fs").encode()))
        self.n_bins = None
        self.rep = 2

    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError(self.lib.gh_error().decode())
        return rc

    def scene(self, arrays):
        self.arrays = arrays
        bvh, bw, bh = pad_bvh(arrays.bvh)
        tri, tw, th = pad_buffer(arrays.tri, 3, 3)
        mat, mw, mh = pad_buffer(arrays.mat, 4, 3)
        norm, nw, nh = pad_buffer(arrays.norm, 9, 3)
        uv, uw, uh = pad_buffer(arrays.uv, 3, 2)
        fp = lambda a: a.ctypes.data_as(_F)
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8))
        self._keep = (bvh, tri, mat, norm, uv)
        self._ck(self.lib.gh_scene(fp(bvh), bw, bh, fp(tri), tw, th, fp(mat), mw, mh, fp(norm), nw, nh, fp(uv), uw, uh,
                                   u8(arrays.atlas), arrays.atlas_res, arrays.atlas_layers,
                                   u8(arrays.env), arrays.env_w, arrays.env_h))
        self.n_bins = arrays.bins.size // 4
        self.bins = np.ascontiguousarray(arrays.bins, dtype=np.uint32)

    def target(self, W, H, replicate=True):
        self.W, self.H = W, H
        self.rep = 2 if replicate else 1
        self._ck(self.lib.gh_target(W, H, self.rep))
        self._ck(self.lib.gh_clear())

    def tracer(self, num_bounces=None, main_override=None, leaf_size=4, post=None):
        src = tracer_source(self.n_bins, leaf_size, self.rep == 2, num_bounces, main_override, post)
        self._ck(self.lib.gh_tracer_program(read_shader("tracer.vs").encode(), src.encode()))

    def tracer_test(self, leaf_size=4):
        src = tracer_test_source(self.n_bins, leaf_size, self.rep == 2)
        self._ck(self.lib.gh_tracer_program(read_shader("tracer.vs").encode(), src.encode()))

    def set_int(self, name, value):
        self._ck(self.lib.gh_set_int(name.encode(), int(value)))

    def clear(self):
        self._ck(self.lib.gh_clear())

    def draw_camera(self, P, I, fov_scale, lens, rand_base):
        self._ck(self.lib.gh_draw_camera((C.c_float * 3)(*P), (C.c_float * 3)(*I), C.c_float(fov_scale),
                                         (C.c_float * 2)(*lens), C.c_float(rand_base)))

    def read_camera(self):
        pos = np.zeros((self.H, self.W, 4), np.float32); d = np.zeros((self.H, self.W, 4), np.float32)
        self._ck(self.lib.gh_read_camera(pos.ctypes.data_as(_F), d.ctypes.data_as(_F)))
        return pos, d

    def set_camera(self, pos, d):
        pos = np.ascontiguousarray(pos, np.float32); d = np.ascontiguousarray(d, np.float32)
        self._ck(self.lib.gh_set_camera(pos.ctypes.data_as(_F), d.ctypes.data_as(_F)))

    def draw_tracer(self, tick, rand_base, env_theta):
        self._ck(self.lib.gh_draw_tracer(C.c_uint(tick), C.c_float(rand_base), C.c_float(env_theta),
                                         self.bins.ctypes.data_as(C.POINTER(C.c_uint)), self.n_bins))

    def read_screen(self, which):
        out = np.zeros((self.H, self.W, 4), np.float32)
        mism = self._ck(self.lib.gh_read_screen(which % 2, out.ctypes.data_as(_F)))
        return out, mism

    def draw(self, acc, exposure=1.0, saturation=1.0, denoise=False, max_sigma=3.0, scale=1.0):
        """drawQuad (main.js:809-824): the reference's draw.fs on an RGBA32F buffer -> RGBA8."""
        if not getattr(self, "_draw_ready", False):
            self._ck(self.lib.gh_draw_program(read_shader("draw.vs").encode(), read_shader("draw.fs").encode()))
            self._draw_ready = True
        acc = np.ascontiguousarray(acc, np.float32)
        H, W = acc.shape[:2]
        out = np.zeros((H, W, 4), np.uint8)
        self._ck(self.lib.gh_draw(acc.ctypes.data_as(_F), W, H, C.c_float(exposure), C.c_float(saturation), C.c_float(scale),
                                  C.c_float(max_sigma), 1 if denoise else 0, out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def write_texture(self, rgba, res, corrected=False, swizzle=(0, 1, 2, 3)):
        """WebGLTextureWriter (texture_packer.js:66-185): the writer's shader strings are taken from the
        reference file at run time."""
        if not getattr(self, "_writer_ready", False):
            js = open(os.path.join(REF, "texture_packer.js")).read()
            vs = re.search(r"let vsStr = `(.*?)`;", js, re.S).group(1)
            fs = re.search(r"let fsStr = `(.*?)`;", js, re.S).group(1)
            self._ck(self.lib.gh_writer_program(vs.encode(), fs.encode()))
            self._writer_ready = True
        rgba = np.ascontiguousarray(rgba, np.uint8)
        h, w = rgba.shape[:2]
        out = np.zeros((res, res, 4), np.uint8)
        sw = (C.c_uint * 4)(*swizzle)
        self._ck(self.lib.gh_write_texture(rgba.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, 1 if corrected else 0, sw, res,
                                           out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out
