'use strict';
/**
 * fspt.js — JavaScript host of libfspt (Node, CommonJS).
 *
 * Mirrors the reference's own host code so that main.js can switch from WebGL2
 * to the MI355X kernels with the same call order (INTEGRATION.md):
 *   buildScene's material step    drives the reference's own TexturePacker / getMaterial / ParseMaterials when the
 *                                 caller passes them (opts.host), a table-driven resolver with the same layer
 *                                 numbering otherwise (AtlasLayers / readMtl / resolveMaterial; images: decoded RGBA8)
 *   packReferenceScene            main.js:355-392 + maskBVHBuffer 272-282, fed with the
 *                                 reference's own BVH / Triangle objects (bvh.js, obj_loader.js)
 *   buildScene                    native obj_loader.js + bvh.js (same decisions, float64) for
 *                                 scenes too large for the JS builder
 *   PathTracer                    drawCamera / drawTracer / tick / clear (main.js:741-857)
 * All compute happens in the HIP kernels behind fspt_napi.node; errors surface as JS Errors.
 */
const addon = require('./fspt_napi.node');
const ABI_VERSION = 4;   // include/fspt.h FSPT_ABI_VERSION this module was written against
if (addon.abiVersion() !== ABI_VERSION) {
  throw new Error('libfspt.so has ABI version ' + addon.abiVersion() + ', fspt.js needs ' + ABI_VERSION + ': stale build (python -c "import __graft_entry__ as g; g.build()")');
}

function jsNum(v) { return String(Number(v)); }

const f = Math.fround;

/** WebGLTextureWriter.setAndDrawTexture + its shader (texture_packer.js:103-121,159-175) on the CPU, float32:
 *  bilinear (S = REPEAT, T = CLAMP_TO_EDGE) resample of a decoded image {width, height, data: RGBA8, row 0 = top}
 *  to res x res at uv = (x+.5, res-(y+.5))/res, sRGB -> linear before filtering when `corrected`, channel swizzle,
 *  rgb premultiplied by alpha, alpha = 1.  Output row 0 = bottom (readPixels).  Same arithmetic as scene.py. */
function resampleImage(img, res, corrected, swizzle) {
  const w = img.width, h = img.height, sw = swizzle || [0, 1, 2, 3];
  const lut = new Float32Array(256), lin = new Float32Array(256);
  for (let i = 0; i < 256; i++) {
    const c = f(i / 255);
    lin[i] = c;
    lut[i] = c <= 0.04045 ? f(c / 12.92) : f(Math.pow(f(f(c + 0.055) / 1.055), 2.4));
  }
  const out = new Uint8Array(res * res * 4);
  const i0 = new Int32Array(res), i1 = new Int32Array(res), ax = new Float32Array(res);
  for (let x = 0; x < res; x++) {
    const px = f(f(x + 0.5) / res), u = f(f(px * w) - 0.5), fl = Math.floor(u);
    ax[x] = f(u - fl);
    i0[x] = ((fl % w) + w) % w; i1[x] = (((fl + 1) % w) + w) % w;
  }
  const c = new Float32Array(4);
  for (let y = 0; y < res; y++) {
    const py = f(f(y + 0.5) / res), v = f(f(f(1 - py) * h) - 0.5), fl = Math.floor(v), by = f(v - fl);
    const j0 = Math.min(Math.max(fl, 0), h - 1), j1 = Math.min(Math.max(fl + 1, 0), h - 1);
    for (let x = 0; x < res; x++) {
      const a = ax[x], na = f(1 - a), nb = f(1 - by);
      for (let k = 0; k < 4; k++) {
        const t = (k < 3 && corrected) ? lut : lin;
        const p00 = t[img.data[(j0 * w + i0[x]) * 4 + k]], p01 = t[img.data[(j0 * w + i1[x]) * 4 + k]];
        const p10 = t[img.data[(j1 * w + i0[x]) * 4 + k]], p11 = t[img.data[(j1 * w + i1[x]) * 4 + k]];
        const top = f(f(p00 * na) + f(p01 * a)), bot = f(f(p10 * na) + f(p11 * a));
        c[k] = f(f(top * nb) + f(bot * by));
      }
      const al = c[sw[3]], o = (y * res + x) * 4;
      for (let k = 0; k < 3; k++) {
        const q = Math.floor(f(f(f(c[sw[k]] * al) * 255) + 0.5));
        out[o + k] = Math.min(Math.max(q, 0), 255);
      }
      out[o + 3] = 255;
    }
  }
  return out;
}

/* ---------------------------------------------------------------------------------------------------------------
 * Material resolution for buildScene.  In the reference this is host code that STAYS in the reference
 * (INTEGRATION.md 2: TexturePacker texture_packer.js:5-63, ParseMaterials mtl_loader.js:3-43, getMaterial
 * main.js:206-270): a caller that has those modules passes them in (`opts.host = {TexturePacker, getMaterial,
 * ParseMaterials}`) and buildScene drives them unchanged.  Without them - Node scripts, the tests on the GPU box -
 * the table-driven resolver below gives the same layer numbering (pinned to the reference's JS by the 'mtl' golden,
 * tests/test_node_host.py::test_js_full_scene_build_matches_reference_js).
 * ------------------------------------------------------------------------------------------------------------- */

/** The atlas layer list: one entry per distinct image (by URL) or flat colour (by its components), in first-use
 *  order.  Like the reference's packer an entry that landed at index 0 is never found again (its lookup tests the
 *  stored index for truthiness), so the first layer may appear twice - the layer ids in `mat` depend on it. */
class AtlasLayers {
  constructor(limit) { this.res = limit || 2048; this.entries = []; this.index = new Map(); this.tallest = 1; }
  _intern(key, entry) {
    const at = this.index.get(key);
    if (at) return at;
    this.entries.push(entry);
    this.index.set(key, this.entries.length - 1);
    return this.entries.length - 1;
  }
  image(img, corrected) {
    const known = this.index.get(img.currentSrc);
    if (known) return known;
    if (img.height > this.tallest) this.tallest = img.height;
    img.corrected = corrected;
    return this._intern(img.currentSrc, img);
  }
  colour(rgb) { return this._intern(rgb.join(' '), rgb); }
  resolution() { this.res = Math.min(this.res, this.tallest); return this.res; }
  /** RGBA8 texels of every layer, res x res each: flat colours as gl.clearColor + readPixels would store them, images
   *  through resampleImage */
  pixels() {
    const res = this.resolution(), n = res * res * 4;
    const out = new Uint8Array(n * this.entries.length);
    this.entries.forEach((e, i) => {
      if (!Array.isArray(e)) { out.set(resampleImage(e, res, !!e.corrected, e.swizzle), i * n); return; }
      const px = [0, 1, 2].map((k) => Math.floor(Math.min(Math.max(Number(e[k]), 0), 1) * 255 + 0.5));
      for (let t = i * n; t < (i + 1) * n; t += 4) { out[t] = px[0]; out[t + 1] = px[1]; out[t + 2] = px[2]; out[t + 3] = 255; }
    });
    return out;
  }
  describe() {
    return this.entries.map((e) => (Array.isArray(e) ? { color: e.map(Number) } :
      { src: e.currentSrc, corrected: !!e.corrected, swizzle: e.swizzle ? Array.from(e.swizzle, Number) : null }));
  }
}

/** MTL statements the pipeline reads: s = one number, v = a list of numbers, u = a file name relative to the OBJ. */
const MTL_FIELDS = { ns: 's', ni: 's', d: 's', illum: 's', dielectric: 's', ior: 's',
  ka: 'v', kd: 'v', kem: 'v', ks: 'v', ke: 'v', pr: 'v', pm: 'v', pmr: 'v', pmr_swizzle: 'v',
  map_bump: 'u', map_kd: 'u', map_kem: 'u', map_ks: 'u', map_d: 'u', map_ns: 'u', map_pmr: 'u' };
/** MTL text -> {materials: {name: {field: value}}, urls: Set}.  Statements before the first `newmtl` are ignored; a
 *  statement whose value is falsy (a scalar 0, an unparsable number) is dropped, as in the reference's loader. */
function readMtl(text, basePath) {
  const materials = {}, urls = new Set();
  let cur = null;
  for (const raw of text.split('\n')) {
    const [word, ...rest] = raw.trim().split(/[ ]+/);
    const field = word.toLowerCase();
    if (field === 'newmtl') { cur = materials[rest[0]] = {}; continue; }
    const kind = MTL_FIELDS[field];
    if (!cur || !kind) continue;
    const value = kind === 's' ? parseFloat(rest[0]) : (kind === 'v' ? rest.map(parseFloat) : rest[0]);
    if (!value) continue;
    if (kind === 'u') urls.add(basePath + '/' + value);
    cur[field] = value;
  }
  return { materials, urls };
}

/** The four atlas layers of a material, in the order their ids are handed out.  Per layer, first match wins:
 *  the group's MTL image map, the group's MTL colour, the prop's scene-JSON field (an image URL, or - where `propColour` -
 *  a colour), the default colour. */
const MATERIAL_LAYERS = [
  { id: 'diffuseIndex', map: 'map_kd', colour: 'kd', prop: 'diffuse', propColour: true, fallback: [0.5, 0.5, 0.5], corrected: true },
  { id: 'roughnessIndex', map: 'map_pmr', colour: 'pmr', prop: 'metallicRoughness', propColour: true, fallback: [0.0, 0.3, 0],
    mapSwizzle: 'pmr_swizzle', propSwizzle: 'mrSwizzle' },
  { id: 'specularIndex', map: 'map_kem', colour: 'kem', prop: 'emission', propColour: false, fallback: [0, 0, 0] },
  { id: 'normalIndex', map: 'map_bump', colour: null, prop: 'normal', propColour: false, anyTruthyProp: true, fallback: [0.5, 0.5, 1] },
];
function resolveMaterial(prop, group, layers, assets, basePath) {
  const mtl = (group && group.material) || {};
  const decoded = (url) => {
    const img = assets && assets[url];
    if (!img) throw new Error('texture ' + url + ' is not in assets');
    if (!img.currentSrc) img.currentSrc = url;
    return img;
  };
  const out = {};
  for (const L of MATERIAL_LAYERS) {
    const pv = prop[L.prop];
    let img = null, swizzle;
    if (mtl[L.map]) { img = decoded(basePath + '/' + mtl[L.map]); swizzle = mtl[L.mapSwizzle]; }
    else if (L.colour && mtl[L.colour]) { out[L.id] = layers.colour(mtl[L.colour]); continue; }
    else if (typeof pv === 'string' || (L.anyTruthyProp && pv)) { img = decoded(pv); swizzle = prop[L.propSwizzle]; }
    else if (L.propColour && pv && typeof pv === 'object') { out[L.id] = layers.colour(pv); continue; }
    else { out[L.id] = layers.colour(L.fallback); continue; }
    // the swizzle lives on the shared decoded image and is (re)set before de-duplication: the last use decides
    if (L.mapSwizzle) img.swizzle = swizzle;
    out[L.id] = L.corrected ? layers.image(img, true) : layers.image(img);
  }
  out.ior = Number(mtl.ior || prop.ior || 1.4);
  out.dielectric = Number(mtl.dielectric || prop.dielectric || -1);
  out.emittance = prop.emittance || [0, 0, 0];
  return out;
}

/** mergeSceneProps (main.js:869-871) */
function mergeSceneProps(scene) {
  return [].concat((scene.props || []), (scene.static_props || []), Object.values(scene.animated_props || []));
}

/** main.js:355-392 for a reference `BVH` instance whose triangles carry `.material`. */
function packReferenceScene(bvh) {
  const bvhArray = bvh.serializeTree();
  const bvhBuffer = [], tri = [], mat = [], norm = [], uv = [];
  for (let i = 0; i < bvhArray.length; i++) {
    const e = bvhArray[i], node = e.node;
    const triIndex = node.leaf ? tri.length / 9 : -1;
    if (node.leaf) {
      for (const t of node.getTriangles()) {
        tri.push(...t.verts[0], ...t.verts[1], ...t.verts[2]);
        const m = t.material;
        mat.push(m.diffuseIndex, m.specularIndex, m.normalIndex, m.roughnessIndex, 0, 0, ...m.emittance, m.ior, m.dielectric, 0);
        for (let k = 0; k < 3; k++) norm.push(...t.normals[k], ...t.tangents[k], ...t.bitangents[k]);
        uv.push(...t.uvs[0], ...t.uvs[1], ...t.uvs[2]);
      }
    }
    bvhBuffer.push(e.left, e.right, triIndex, ...node.boundingBox.min, ...node.boundingBox.max);
  }
  const masked = new Float32Array(new Int32Array(bvhBuffer).buffer);   // maskBVHBuffer
  for (let i = 0; i < bvhBuffer.length; i += 9) for (let j = 3; j < 9; j++) masked[i + j] = bvhBuffer[i + j];
  return { bvh: masked, tri: new Float32Array(tri), mat: new Float32Array(mat), norm: new Float32Array(norm),
           uv: new Float32Array(uv), depth: bvh.depth };
}

/** initBVH (main.js:284-445) through the native builder (same decisions, float64).
 *  sceneOrProps: the scene JSON (props / static_props / animated_props, worldTransforms, normalize, atlasRes) or a
 *  bare props array; objTexts: {path: OBJ text}; env: {rgbe, width, height} | null;
 *  opts: {mtlTexts: {url: MTL text}, assets: {url: decoded image}, focusRays: [[eye, dir], ...],
 *         host: {TexturePacker, getMaterial, ParseMaterials} (the reference's modules) + atlasPixels(packer)}. */
function buildScene(sceneOrProps, objTexts, env, leafSize, opts) {
  opts = opts || {};
  const scene = Array.isArray(sceneOrProps) ? { props: sceneOrProps } : sceneOrProps;
  const props = mergeSceneProps(scene);
  // the reference's own host modules when the caller has them (INTEGRATION.md), the resolver above otherwise
  const host = opts.host || null;
  const packer = host ? new host.TexturePacker(scene.atlasRes || 2048, props.length) : new AtlasLayers(scene.atlasRes || 2048);
  const mtlOf = host ? host.ParseMaterials : readMtl;
  const materialOf = host ? host.getMaterial : resolveMaterial;
  const b = addon.builderCreate();
  let s;
  try {
    for (const p of props) {
      const basePath = p.path.split('/').slice(0, -1).join('/');
      const groups = addon.builderParseObj(b, objTexts[p.path], { rotate: p.rotate || [], scale: p.scale === undefined ? 1 : p.scale,
        translate: p.translate || [0, 0, 0], normals: p.normals || 'flat' }, scene.worldTransforms || null, p.skips || null);
      const libs = {};
      const mats = groups.map((g) => {
        let material = {};
        if (g.mtllib !== null) {
          const url = basePath + '/' + g.mtllib;
          if (!libs[url]) {
            if (!opts.mtlTexts || opts.mtlTexts[url] === undefined) throw new Error('mtllib ' + url + ' is not in opts.mtlTexts');
            libs[url] = mtlOf(opts.mtlTexts[url], basePath).materials;
          }
          material = libs[url][g.name] || {};
        }
        return materialOf(p, { material }, packer, opts.assets, basePath);
      });
      addon.builderCommit(b, mats);
    }
    if (scene.normalize) addon.builderNormalize(b, scene.normalize);
    s = addon.builderBuild(b, leafSize || 4);
    s.focus = (opts.focusRays || []).map(([eye, dir]) => 1 - 1 / addon.builderAutofocus(b, eye, dir));   // main.js:544
  } finally {
    addon.builderDestroy(b);
  }
  if (host) {
    // the reference's packer renders the atlas with a WebGL context of its own (texture_packer.js:103-175): its pixels
    // are the caller's to fetch (opts.atlasPixels(packer) -> Uint8Array of res*res*4*layers); the layer ids are in `mat`
    if (typeof opts.atlasPixels !== 'function') throw new Error('buildScene: opts.host needs opts.atlasPixels(packer)');
    s.atlasRes = packer.setAndGetResolution(); s.atlasLayers = packer.imageSet.length;
    s.atlas = opts.atlasPixels(packer);
    s.layers = null;
  } else {
    s.atlas = packer.pixels(); s.atlasRes = packer.res; s.atlasLayers = packer.entries.length;
    s.layers = packer.describe();
  }
  if (env) { s.env = env.rgbe; s.envW = env.width; s.envH = env.height; s.bins = addon.envBins(env.rgbe, env.width, env.height); }
  else { s.env = null; s.envW = 0; s.envH = 0; s.bins = new Uint32Array([0, 0, 1, 2048]); }   // main.js:292
  s.leafSize = leafSize || 4;
  return s;
}

/** `.fspt` scene blob (layout: fspt_amd/blob.py). */
const fs = require('fs');
const DT = [Float32Array, Uint8Array, Uint32Array];
function saveBlob(path, s) {
  const secs = [['bvh', s.bvh], ['tri', s.tri], ['mat', s.mat], ['norm', s.norm], ['uv', s.uv], ['atlas', s.atlas],
    ['bins', s.bins], ['meta', new Uint32Array([s.atlasRes, s.atlasLayers, s.env ? s.envW : 0, s.env ? s.envH : 0, s.leafSize, s.depth || 0])]];
  if (s.env) secs.push(['env', s.env]);
  const parts = [Buffer.from('FSPT'), Buffer.alloc(8)];
  parts[1].writeUInt32LE(1, 0); parts[1].writeUInt32LE(secs.length, 4);
  for (const [name, arr] of secs) {
    const h = Buffer.alloc(24);
    h.write(name, 0, 'ascii');
    h.writeUInt32LE(DT.findIndex((T) => arr instanceof T), 8);
    h.writeBigUInt64LE(BigInt(arr.length), 16);
    const raw = Buffer.from(arr.buffer, arr.byteOffset, arr.byteLength);
    parts.push(h, raw, Buffer.alloc((16 - (raw.length % 16)) % 16));
  }
  fs.writeFileSync(path, Buffer.concat(parts));
}
function loadBlob(path) {
  const buf = fs.readFileSync(path);
  if (buf.toString('ascii', 0, 4) !== 'FSPT' || buf.readUInt32LE(4) !== 1) throw new Error('not a version-1 .fspt blob');
  const n = buf.readUInt32LE(8), secs = {};
  let off = 12;
  for (let i = 0; i < n; i++) {
    const name = buf.toString('ascii', off, off + 8).replace(/\0+$/, '');
    const T = DT[buf.readUInt32LE(off + 8)], count = Number(buf.readBigUInt64LE(off + 16));
    off += 24;
    const nbytes = count * T.BYTES_PER_ELEMENT;
    secs[name] = new T(buf.buffer.slice(buf.byteOffset + off, buf.byteOffset + off + nbytes));
    off += nbytes + ((16 - (nbytes % 16)) % 16);
  }
  const m = secs.meta;
  return { bvh: secs.bvh, tri: secs.tri, mat: secs.mat, norm: secs.norm, uv: secs.uv, atlas: secs.atlas, atlasRes: m[0],
    atlasLayers: m[1], env: secs.env || null, envW: m[2], envH: m[3], bins: secs.bins, leafSize: m[4], depth: m[5] };
}

const PIPELINE_CODES = { megakernel: 0, wavefront: 1, stream: 2 };

// a pipeline name (or a code of include/fspt_tuning.h's fspt_target_set_pipeline); anything else is an error, as in the Python host
function pipelineCode(name) {
  if (PIPELINE_CODES[name] !== undefined) return PIPELINE_CODES[name];
  if (Number.isInteger(name) && name >= 0 && name <= 2) return name;
  throw new Error("unknown pipeline '" + name + "' (want " + Object.keys(PIPELINE_CODES).join(', ') + ')');
}

class PathTracer {
  /** scene: {bvh,tri,mat,norm,uv,atlas,atlasRes,atlasLayers,env,envW,envH,bins,leafSize} */
  constructor(scene, width, height, device) {
    this.resolution = [width, height];
    this._scene = addon.sceneCreate(scene, device || 0);
    this._target = addon.targetCreate(this._scene, width, height);
    // main.js:67-74
    this.fovScale = 0.5; this.envTheta = 0; this.dir = [0, 0, -1]; this.eye = [0, 0, 2];
    this.lensFeatures = [1 - 1 / 2.0, 0.02];
    this.numBounces = 4;                       // tracer.fs:9
    this.pingpong = 0;
    this._rng = new BigUint64Array([1n]);      // replaces Math.random()*10000 (main.js:748,777)
  }
  seed(s) { this._rng[0] = BigInt(s); }
  _randBase() { return addon.randBaseNext(this._rng); }
  drawCamera(randBase) { addon.camera(this._target, this.eye, this.dir, this.fovScale, this.lensFeatures, randBase === undefined ? this._randBase() : randBase); }
  drawTracer(i, randBase) { addon.trace(this._target, i, randBase === undefined ? this._randBase() : randBase, this.envTheta, this.numBounces); }
  drawTracerTest(i) { addon.traceTest(this._target, i); }   // mode=test: bvh_test.fs (main.js:879-883)
  tick() { this.drawCamera(); this.drawTracer(this.pingpong); this.pingpong++; }            // main.js:838-857
  render(nTicks) {
    addon.render(this._target, { P: this.eye, I: this.dir, fovScale: this.fovScale, lens: this.lensFeatures,
      envTheta: this.envTheta, numBounces: this.numBounces }, this.pingpong, nTicks, this._rng[0]);
    for (let k = 0; k < 2 * nTicks; k++) this._randBase();
    this.pingpong += nTicks;
  }
  /** render(nTicks) on a worker thread (napi_async_work): resolves when the ticks are on the device's accumulator.
   *  Until it settles every other call on this tracer throws Error('render in flight') (enforced by the addon: the
   *  library's contract is one thread at a time per target, like the reference's single-threaded tick(),
   *  main.js:838-857); close() waits for it. */
  renderAsync(nTicks) {
    const p = addon.renderAsync(this._target, { P: this.eye, I: this.dir, fovScale: this.fovScale, lens: this.lensFeatures,
      envTheta: this.envTheta, numBounces: this.numBounces }, this.pingpong, nTicks, this._rng[0]);
    for (let k = 0; k < 2 * nTicks; k++) this._randBase();
    this.pingpong += nTicks;
    const settled = p.then(() => {}, () => {});
    this._inflight = settled;
    settled.then(() => { if (this._inflight === settled) this._inflight = null; });
    return p;
  }
  clear() { addon.clear(this._target); this.pingpong = 0; }                                  // main.js:826-836
  /** wait until every enqueued (and recorded) tick is on the accumulator (gl.finish) */
  sync() { addon.sync(this._target); }
  readRadiance(out) {
    out = out || new Float32Array(this.resolution[0] * this.resolution[1] * 4);
    if (out.length !== this.resolution[0] * this.resolution[1] * 4) throw new RangeError('readRadiance: need W*H*4 floats');
    return addon.readRadiance(this._target, out);
  }
  /** drawQuad (main.js:809-824) -> draw.fs: tonemapped RGBA8 as the canvas would hold it (row 0 = bottom). */
  drawQuad(exposure, saturation, denoise, maxSigma, out, resScale) {
    out = out || new Uint8Array(this.resolution[0] * this.resolution[1] * 4);
    if (out.length !== this.resolution[0] * this.resolution[1] * 4) throw new RangeError('drawQuad: need W*H*4 bytes');
    return addon.draw(this._target, exposure === undefined ? 1 : exposure, saturation === undefined ? 1 : saturation,
      !!denoise, maxSigma === undefined ? 3 : maxSigma, out, resScale === undefined ? 1 : resScale);   // draw.fs `scale`
  }
  /** gl.viewport(0, 0, w, h) of drawCamera / drawTracer (main.js:744,761); the reference: resolution * resScale. */
  setViewport(w, h) { addon.setViewport(this._target, w || 0, h || 0); }
  setShard(shard, nShards, tile) { addon.setShard(this._target, shard, nShards, tile || 32); }
  /** 'wavefront' (batches of ticks), 'stream' (fixed pool of live paths), 'megakernel' (include/fspt_tuning.h) */
  setPipeline(name, batch) { addon.setPipeline(this._target, pipelineCode(name), batch || 0); }
  /** traversal steps a starved trace wave walks on before it suspends its rays (0 = never; include/fspt.h) */
  setTraceBudget(steps) { addon.setTraceBudget(this._target, steps); }
  /** stream scheduler: live paths per state set (0 = default), drain iterations (-1 = default), iteration cap (0 = none) */
  setPool(paths, drain, maxIterations, overlap) { addon.setPool(this._target, paths || 0, drain === undefined ? -1 : drain, maxIterations || 0, overlap === undefined ? -1 : overlap); }
  /** -1 adaptive (default), 0 never, r >= 1: the tail kernel takes the live paths over after wavefront round r */
  setTail(round) { addon.setTail(this._target, round === undefined ? -1 : round); }
  /** drawCamera + drawTracer ticks are recorded and run as batches at the next read-out (default, include/fspt.h);
   *  false: every drawTracer executes at once */
  setDeferred(on) { addon.setDeferred(this._target, on !== false); }
  /** HIP event pairs around every kernel launch (per-kernel timing for measurement hosts; ~1.3 % of a 20-tick batch): on by default */
  setStageTiming(on) { addon.setStageTiming(this._target, on !== false); }
  /** cap / query the wavefront path state (include/fspt.h: fspt_target_set_memory_limit) and pre-allocate it */
  setMemoryLimit(bytes) { addon.setMemoryLimit(this._target, bytes || 0); }
  pathStateBytes() { return addon.pathStateBytes(this._target); }
  prepare() { addon.prepare(this._target); }
  enableCounters(on) { addon.enableCounters(this._target, !!on); }
  counters() { return addon.counters(this._target); }
  /** Recorded (deferred) ticks are executed first: fspt_target_destroy itself drops them (it never writes to a
   *  caller-owned accumulator, include/fspt.h).  With a renderAsync in flight close() waits for it and returns a
   *  Promise; otherwise it is synchronous.  (A tracer that is simply dropped is cleaned up by the handles' finalizers.) */
  close() {
    if (this._inflight) return this._inflight.then(() => this.close());
    if (this._target) { try { addon.sync(this._target); } finally { addon.targetDestroy(this._target); addon.sceneDestroy(this._scene); this._target = null; this._scene = null; } }
    return undefined;
  }
}

/** The same frame driver over several GPUs of the node from this one JS thread (include/fspt.h: fspt_multi_*):
 *  every device traces every devices.length-th 32x32 tile, nothing moves between devices while rendering, and
 *  readRadiance() / drawQuad() gather the tiles onto devices[0] with peer-to-peer copies.  Bit-identical to one GPU. */
class MultiPathTracer {
  constructor(scene, width, height, devices) {
    this.resolution = [width, height];
    this.devices = devices && devices.length ? devices.slice() : [0];
    this._multi = addon.multiCreate(scene, this.devices, width, height);
    this.fovScale = 0.5; this.envTheta = 0; this.dir = [0, 0, -1]; this.eye = [0, 0, 2];
    this.lensFeatures = [1 - 1 / 2.0, 0.02];
    this.numBounces = 4;
    this.pingpong = 0;
    this._rng = new BigUint64Array([1n]);
  }
  seed(s) { this._rng[0] = BigInt(s); }
  _randBase() { return addon.randBaseNext(this._rng); }
  _params() { return { P: this.eye, I: this.dir, fovScale: this.fovScale, lens: this.lensFeatures, envTheta: this.envTheta, numBounces: this.numBounces }; }
  drawCamera(randBase) { addon.multiCamera(this._multi, this.eye, this.dir, this.fovScale, this.lensFeatures, randBase === undefined ? this._randBase() : randBase); }
  drawTracer(i, randBase) { addon.multiTrace(this._multi, i, randBase === undefined ? this._randBase() : randBase, this.envTheta, this.numBounces); }
  tick() { this.drawCamera(); this.drawTracer(this.pingpong); this.pingpong++; }
  _advance(nTicks) { for (let k = 0; k < 2 * nTicks; k++) this._randBase(); this.pingpong += nTicks; }
  render(nTicks) { addon.multiRender(this._multi, this._params(), this.pingpong, nTicks, this._rng[0]); this._advance(nTicks); }
  /** as PathTracer.renderAsync: other calls throw Error('render in flight') until it settles; close() waits */
  renderAsync(nTicks) {
    const p = addon.multiRenderAsync(this._multi, this._params(), this.pingpong, nTicks, this._rng[0]);
    this._advance(nTicks);
    const settled = p.then(() => {}, () => {});
    this._inflight = settled;
    settled.then(() => { if (this._inflight === settled) this._inflight = null; });
    return p;
  }
  /** per device [render, pack, transfer, scatter] ms of the most recent render + read-out (fspt_multi_last_stage_ms; -1 = did not run) */
  stageMs() {
    const flat = addon.multiLastStageMs(this._multi, this.devices.length), out = [];
    for (let i = 0; i < this.devices.length; i++) out.push(Array.from(flat.subarray(4 * i, 4 * i + 4)));
    return out;
  }
  clear() { addon.multiClear(this._multi); this.pingpong = 0; }
  sync() { addon.multiSync(this._multi); }
  /** the read-out exchange (include/fspt_multi.h): 'peer' (hipMemcpyPeerAsync of the packed tiles, default), 'rccl_gather'
   *  (ncclSend / ncclRecv of the same tiles), 'rccl_reduce' (ncclReduce(SUM) of own-tiles-only frames); RCCL needs distinct devices */
  setExchange(mode) {
    const codes = { peer: 0, rccl_gather: 1, rccl_reduce: 2 };
    if (codes[mode] === undefined) throw new Error("unknown exchange '" + mode + "' (want " + Object.keys(codes).join(', ') + ')');
    addon.multiSetExchange(this._multi, codes[mode]);
  }
  exchange() { return addon.multiGetExchange(this._multi); }
  readRadiance(out) {
    out = out || new Float32Array(this.resolution[0] * this.resolution[1] * 4);
    if (out.length !== this.resolution[0] * this.resolution[1] * 4) throw new RangeError('readRadiance: need W*H*4 floats');
    return addon.multiReadRadiance(this._multi, out);
  }
  drawQuad(exposure, saturation, denoise, maxSigma, out) {
    out = out || new Uint8Array(this.resolution[0] * this.resolution[1] * 4);
    if (out.length !== this.resolution[0] * this.resolution[1] * 4) throw new RangeError('drawQuad: need W*H*4 bytes');
    return addon.multiDraw(this._multi, exposure === undefined ? 1 : exposure, saturation === undefined ? 1 : saturation,
      !!denoise, maxSigma === undefined ? 3 : maxSigma, out);
  }
  setPipeline(name, batch) {
    const code = pipelineCode(name);
    for (let i = 0; i < this.devices.length; i++) addon.setPipeline(addon.multiTarget(this._multi, i), code, batch || 0);
  }
  close() {
    if (this._inflight) return this._inflight.then(() => this.close());
    if (this._multi) { addon.multiDestroy(this._multi); this._multi = null; }
    return undefined;
  }
}

// memory later scenes may spend on interleaved material textures (fspt_set_texture_interleave_budget; results do not depend on it)
function setTextureInterleaveBudget(bytes) { addon.setTextureInterleaveBudget(bytes); }

module.exports = { addon, setTextureInterleaveBudget, MultiPathTracer, AtlasLayers, resolveMaterial, readMtl, mergeSceneProps, resampleImage, packReferenceScene, buildScene,
  PathTracer, saveBlob, loadBlob };
