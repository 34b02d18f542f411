'use strict';
/**
 * fspt.js — JavaScript host of libfspt (Node, CommonJS).
 *
 * Mirrors the reference's own host code so that main.js can switch from WebGL2
 * to the MI355X kernels with the same call order (INTEGRATION.md):
 *   TexturePacker / getMaterial   texture_packer.js:5-63, main.js:206-270 (colour-valued maps)
 *   packReferenceScene            main.js:355-392 + maskBVHBuffer 272-282, fed with the
 *                                 reference's own BVH / Triangle objects (bvh.js, obj_loader.js)
 *   buildScene                    native obj_loader.js + bvh.js (same decisions, float64) for
 *                                 scenes too large for the JS builder
 *   PathTracer                    drawCamera / drawTracer / tick / clear (main.js:741-857)
 * All compute happens in the HIP kernels behind fspt_napi.node; errors surface as JS Errors.
 */
const addon = require('./fspt_napi.node');

function jsNum(v) { return String(Number(v)); }

class TexturePacker {
  constructor(atlasRes) { this.res = atlasRes || 2048; this.imageSet = []; this.imageKeys = {}; this.maxRes = 1; }
  addColor(color) {
    const key = color.map(jsNum).join(' ');
    if (this.imageKeys[key]) return this.imageKeys[key];   // index 0 is falsy: never de-duplicated (texture_packer.js:27)
    this.imageSet.push(color.map(Number));
    this.imageKeys[key] = this.imageSet.length - 1;
    return this.imageKeys[key];
  }
  setAndGetResolution() { if (this.maxRes < this.res) this.res = this.maxRes; return this.res; }
  getPixels() {
    const res = this.setAndGetResolution();
    const out = new Uint8Array(res * res * 4 * this.imageSet.length);
    this.imageSet.forEach((c, i) => {
      const px = [0, 1, 2].map((k) => Math.floor(Math.min(Math.max(c[k], 0), 1) * 255 + 0.5)).concat([255]);
      for (let t = 0; t < res * res; t++) out.set(px, (i * res * res + t) * 4);
    });
    return out;
  }
}

function getMaterial(prop, packer) {
  const colour = (v, d) => (Array.isArray(v) ? v : d);
  const material = {};
  material.diffuseIndex = packer.addColor(colour(prop.diffuse, [0.5, 0.5, 0.5]));
  material.roughnessIndex = packer.addColor(colour(prop.metallicRoughness, [0.0, 0.3, 0]));
  material.specularIndex = packer.addColor(colour(prop.emission, [0, 0, 0]));
  material.normalIndex = packer.addColor([0.5, 0.5, 1]);
  material.ior = prop.ior || 1.4;
  material.dielectric = prop.dielectric || -1;
  material.emittance = prop.emittance || [0, 0, 0];
  return material;
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

/** initBVH (main.js:284-445) through the native builder.  props: scene-JSON props, objTexts: {path: text}. */
function buildScene(props, objTexts, env, leafSize) {
  const packer = new TexturePacker();
  const jobs = props.map((p) => ({ obj: objTexts[p.path], rotate: p.rotate || [], scale: p.scale === undefined ? 1 : p.scale,
    translate: p.translate || [0, 0, 0], normals: p.normals || 'flat', material: getMaterial(p, packer) }));
  const s = addon.buildScene(jobs, leafSize || 4);
  s.atlas = packer.getPixels(); s.atlasRes = packer.res; s.atlasLayers = packer.imageSet.length;
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
  tick() { this.drawCamera(); this.drawTracer(this.pingpong); this.pingpong++; }            // main.js:838-857
  render(nTicks) {
    addon.render(this._target, { P: this.eye, I: this.dir, fovScale: this.fovScale, lens: this.lensFeatures,
      envTheta: this.envTheta, numBounces: this.numBounces }, this.pingpong, nTicks, this._rng[0]);
    for (let k = 0; k < 2 * nTicks; k++) this._randBase();
    this.pingpong += nTicks;
  }
  clear() { addon.clear(this._target); this.pingpong = 0; }                                  // main.js:826-836
  readRadiance(out) {
    out = out || new Float32Array(this.resolution[0] * this.resolution[1] * 4);
    if (out.length !== this.resolution[0] * this.resolution[1] * 4) throw new RangeError('readRadiance: need W*H*4 floats');
    return addon.readRadiance(this._target, out);
  }
  /** drawQuad (main.js:809-824) -> draw.fs: tonemapped RGBA8 as the canvas would hold it (row 0 = bottom). */
  drawQuad(exposure, saturation, denoise, maxSigma, out) {
    out = out || new Uint8Array(this.resolution[0] * this.resolution[1] * 4);
    if (out.length !== this.resolution[0] * this.resolution[1] * 4) throw new RangeError('drawQuad: need W*H*4 bytes');
    return addon.draw(this._target, exposure === undefined ? 1 : exposure, saturation === undefined ? 1 : saturation,
      !!denoise, maxSigma === undefined ? 3 : maxSigma, out);
  }
  setShard(shard, nShards, tile) { addon.setShard(this._target, shard, nShards, tile || 32); }
  setPipeline(name, batch) { addon.setPipeline(this._target, name === 'megakernel' ? 0 : (name === 'wavefront2' ? 2 : 1), batch || 0); }
  enableCounters(on) { addon.enableCounters(this._target, !!on); }
  counters() { return addon.counters(this._target); }
  close() { if (this._target) { addon.targetDestroy(this._target); addon.sceneDestroy(this._scene); this._target = null; } }
}

module.exports = { addon, TexturePacker, getMaterial, packReferenceScene, buildScene, PathTracer, saveBlob, loadBlob };
