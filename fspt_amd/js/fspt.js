'use strict';
/**
 * fspt.js — JavaScript host of libfspt (Node, CommonJS).
 *
 * Mirrors the reference's own host code so that main.js can switch from WebGL2
 * to the MI355X kernels with the same call order (INTEGRATION.md):
 *   TexturePacker / getMaterial / parseMaterials / mergeSceneProps
 *                                 texture_packer.js:5-63, main.js:206-270,869-871, mtl_loader.js (images: decoded RGBA8)
 *   packReferenceScene            main.js:355-392 + maskBVHBuffer 272-282, fed with the
 *                                 reference's own BVH / Triangle objects (bvh.js, obj_loader.js)
 *   buildScene                    native obj_loader.js + bvh.js (same decisions, float64) for
 *                                 scenes too large for the JS builder
 *   PathTracer                    drawCamera / drawTracer / tick / clear (main.js:741-857)
 * All compute happens in the HIP kernels behind fspt_napi.node; errors surface as JS Errors.
 */
const addon = require('./fspt_napi.node');

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

/** texture_packer.js:5-63; images are decoded RGBA8 objects {currentSrc, width, height, data} instead of
 *  HTMLImageElements, and the atlas is written on the CPU (resampleImage) instead of by a WebGL context. */
class TexturePacker {
  constructor(atlasRes) { this.res = atlasRes || 2048; this.imageSet = []; this.imageKeys = {}; this.maxRes = 1; }
  addTexture(image, corrected) {
    if (this.imageKeys[image.currentSrc]) return this.imageKeys[image.currentSrc];
    this.maxRes = Math.max(this.maxRes, image.height);
    image.corrected = corrected;
    this.imageSet.push(image);
    this.imageKeys[image.currentSrc] = this.imageSet.length - 1;
    return this.imageKeys[image.currentSrc];
  }
  addColor(color) {
    const key = color.join(' ');
    if (this.imageKeys[key]) return this.imageKeys[key];   // index 0 is falsy: never de-duplicated (texture_packer.js:27)
    this.imageSet.push(color);
    this.imageKeys[key] = this.imageSet.length - 1;
    return this.imageKeys[key];
  }
  setAndGetResolution() { if (this.maxRes < this.res) this.res = this.maxRes; return this.res; }
  getPixels() {
    const res = this.setAndGetResolution();
    const out = new Uint8Array(res * res * 4 * this.imageSet.length);
    this.imageSet.forEach((c, i) => {
      if (Array.isArray(c)) {   // gl.clearColor(c) + readPixels RGBA8 (texture_packer.js:152-157)
        const px = [0, 1, 2].map((k) => Math.floor(Math.min(Math.max(Number(c[k]), 0), 1) * 255 + 0.5)).concat([255]);
        for (let t = 0; t < res * res; t++) out.set(px, (i * res * res + t) * 4);
      } else {
        out.set(resampleImage(c, res, !!c.corrected, c.swizzle), i * res * res * 4);
      }
    });
    return out;
  }
  describe() {
    return this.imageSet.map((e) => (Array.isArray(e) ? { color: e.map(Number) } :
      { src: e.currentSrc, corrected: !!e.corrected, swizzle: e.swizzle ? Array.from(e.swizzle, Number) : null }));
  }
}

/** ParseMaterials (mtl_loader.js:3-43). */
function parseMaterials(mtlText, basePath) {
  const materials = {}, urls = new Set();
  const scalarTokens = new Set(['ns', 'ni', 'd', 'illum', 'dielectric', 'ior']);
  const vectorTokens = new Set(['ka', 'kd', 'kem', 'ks', 'ke', 'pr', 'pm', 'pmr', 'pmr_swizzle']);
  const stringTokens = new Set(['map_bump', 'map_kd', 'map_kem', 'map_ks', 'map_d', 'map_ns', 'map_pmr']);
  let mtlName = null;
  for (const line of mtlText.split('\n')) {
    const tokens = line.trim().split(/[ ]+/), key = tokens[0].toLowerCase();
    if (key === 'newmtl') { mtlName = tokens[1]; materials[mtlName] = {}; }
    if (!mtlName) continue;
    let value, isUrl = false;
    if (scalarTokens.has(key)) value = parseFloat(tokens[1]);
    else if (vectorTokens.has(key)) value = tokens.slice(1).map(parseFloat);
    else if (stringTokens.has(key)) { value = tokens[1]; isUrl = true; }
    if (value) {
      if (isUrl) urls.add(basePath + '/' + value);
      materials[mtlName][key] = value;
    }
  }
  return { materials, urls };
}

/** getMaterial (main.js:206-270), same signature: the OBJ group's MTL entry wins over the prop's scene-JSON fields;
 *  assets = {url: decoded image}. */
function getMaterial(transforms, group, texturePacker, assets, basePath) {
  const gm = (group && group.material) || {};
  const asset = (url) => {
    if (!assets || !assets[url]) throw new Error('texture ' + url + ' is not in assets');
    if (!assets[url].currentSrc) assets[url].currentSrc = url;
    return assets[url];
  };
  const material = {};
  if (gm.map_kd) material.diffuseIndex = texturePacker.addTexture(asset(basePath + '/' + gm.map_kd), true);
  else if (gm.kd) material.diffuseIndex = texturePacker.addColor(gm.kd);
  else if (typeof transforms.diffuse === 'string') material.diffuseIndex = texturePacker.addTexture(asset(transforms.diffuse), true);
  else if (typeof transforms.diffuse === 'object' && transforms.diffuse) material.diffuseIndex = texturePacker.addColor(transforms.diffuse);
  else material.diffuseIndex = texturePacker.addColor([0.5, 0.5, 0.5]);

  if (gm.map_pmr) {
    const img = asset(basePath + '/' + gm.map_pmr);
    img.swizzle = gm.pmr_swizzle;            // set on the shared image before de-duplication: the last use decides
    material.roughnessIndex = texturePacker.addTexture(img);
  } else if (gm.pmr) material.roughnessIndex = texturePacker.addColor(gm.pmr);
  else if (typeof transforms.metallicRoughness === 'string') {
    const img = asset(transforms.metallicRoughness);
    img.swizzle = transforms.mrSwizzle;
    material.roughnessIndex = texturePacker.addTexture(img);
  } else if (typeof transforms.metallicRoughness === 'object' && transforms.metallicRoughness) {
    material.roughnessIndex = texturePacker.addColor(transforms.metallicRoughness);
  } else material.roughnessIndex = texturePacker.addColor([0.0, 0.3, 0]);

  if (gm.map_kem) material.specularIndex = texturePacker.addTexture(asset(basePath + '/' + gm.map_kem));
  else if (gm.kem) material.specularIndex = texturePacker.addColor(gm.kem);
  else if (typeof transforms.emission === 'string') material.specularIndex = texturePacker.addTexture(asset(transforms.emission));
  else material.specularIndex = texturePacker.addColor([0, 0, 0]);

  if (gm.map_bump) material.normalIndex = texturePacker.addTexture(asset(basePath + '/' + gm.map_bump));
  else if (transforms.normal) material.normalIndex = texturePacker.addTexture(asset(transforms.normal));
  else material.normalIndex = texturePacker.addColor([0.5, 0.5, 1]);
  material.ior = Number(gm.ior || transforms.ior || 1.4);
  material.dielectric = Number(gm.dielectric || transforms.dielectric || -1);
  material.emittance = transforms.emittance || [0, 0, 0];
  return material;
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
 *  opts: {mtlTexts: {url: MTL text}, assets: {url: decoded image}, focusRays: [[eye, dir], ...]}. */
function buildScene(sceneOrProps, objTexts, env, leafSize, opts) {
  opts = opts || {};
  const scene = Array.isArray(sceneOrProps) ? { props: sceneOrProps } : sceneOrProps;
  const props = mergeSceneProps(scene);
  const packer = new TexturePacker(scene.atlasRes || 2048);
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
            libs[url] = parseMaterials(opts.mtlTexts[url], basePath).materials;
          }
          material = libs[url][g.name] || {};
        }
        return getMaterial(p, { material }, packer, opts.assets, basePath);
      });
      addon.builderCommit(b, mats);
    }
    if (scene.normalize) addon.builderNormalize(b, scene.normalize);
    s = addon.builderBuild(b, leafSize || 4);
    s.focus = (opts.focusRays || []).map(([eye, dir]) => 1 - 1 / addon.builderAutofocus(b, eye, dir));   // main.js:544
  } finally {
    addon.builderDestroy(b);
  }
  s.atlas = packer.getPixels(); s.atlasRes = packer.res; s.atlasLayers = packer.imageSet.length;
  s.layers = packer.describe();
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
   *  Do not call anything else on this tracer until it settles. */
  renderAsync(nTicks) {
    const p = addon.renderAsync(this._target, { P: this.eye, I: this.dir, fovScale: this.fovScale, lens: this.lensFeatures,
      envTheta: this.envTheta, numBounces: this.numBounces }, this.pingpong, nTicks, this._rng[0]);
    for (let k = 0; k < 2 * nTicks; k++) this._randBase();
    this.pingpong += nTicks;
    return p;
  }
  clear() { addon.clear(this._target); this.pingpong = 0; }                                  // main.js:826-836
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
  setPipeline(name, batch) { addon.setPipeline(this._target, name === 'megakernel' ? 0 : (name === 'wavefront2' ? 2 : 1), batch || 0); }
  /** -1 adaptive (default), 0 never, r >= 1: the tail kernel takes the live paths over after wavefront round r */
  setTail(round) { addon.setTail(this._target, round === undefined ? -1 : round); }
  /** drawCamera + drawTracer ticks are recorded and run as batches at the next read-out (default, include/fspt.h);
   *  false: every drawTracer executes at once */
  setDeferred(on) { addon.setDeferred(this._target, on !== false); }
  /** cap / query the wavefront path state (include/fspt.h: fspt_target_set_memory_limit) and pre-allocate it */
  setMemoryLimit(bytes) { addon.setMemoryLimit(this._target, bytes || 0); }
  pathStateBytes() { return addon.pathStateBytes(this._target); }
  prepare() { addon.prepare(this._target); }
  enableCounters(on) { addon.enableCounters(this._target, !!on); }
  counters() { return addon.counters(this._target); }
  close() { if (this._target) { addon.targetDestroy(this._target); addon.sceneDestroy(this._scene); this._target = null; } }
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
  renderAsync(nTicks) { const p = addon.multiRenderAsync(this._multi, this._params(), this.pingpong, nTicks, this._rng[0]); this._advance(nTicks); return p; }
  clear() { addon.multiClear(this._multi); this.pingpong = 0; }
  sync() { addon.multiSync(this._multi); }
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
    const code = name === 'megakernel' ? 0 : (name === 'wavefront2' ? 2 : 1);
    for (let i = 0; i < this.devices.length; i++) addon.setPipeline(addon.multiTarget(this._multi, i), code, batch || 0);
  }
  close() { if (this._multi) { addon.multiDestroy(this._multi); this._multi = null; } }
}

// memory later scenes may spend on interleaved material textures (fspt_set_texture_interleave_budget; results do not depend on it)
function setTextureInterleaveBudget(bytes) { addon.setTextureInterleaveBudget(bytes); }

module.exports = { addon, setTextureInterleaveBudget, MultiPathTracer, TexturePacker, getMaterial, parseMaterials, mergeSceneProps, resampleImage, packReferenceScene, buildScene,
  PathTracer, saveBlob, loadBlob };
