'use strict';
// Driven by tests/test_node_host.py: node node_host_check.js <mode> <job.json> <out.json>
const fs = require('fs');
const path = require('path');
const F = require(path.join(__dirname, '..', 'fspt_amd', 'js', 'fspt.js'));
const mode = process.argv[2];
const job = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));
const out = {};
const b64 = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength).toString('base64');
const env = job.env ? { rgbe: Uint8Array.from(Buffer.from(job.env.rgbe_b64, 'base64')), width: job.env.width, height: job.env.height } : null;
const SF = require(path.join(__dirname, '..', 'fspt_amd', 'js', 'scene_file.js'));
if (mode === 'scene_file') {
  // scene JSON + OBJ / MTL / PNG maps / RGBE sky on disk -> arrays (loadSceneFile: main.js:915-950 + initBVH)
  const { scene: s, settings } = SF.loadSceneFile(job.scene_path);
  for (const k of ['bvh', 'tri', 'mat', 'norm', 'uv', 'bins', 'atlas', 'env']) out[k] = b64(s[k]);
  out.depth = s.depth; out.atlasLayers = s.atlasLayers; out.atlasRes = s.atlasRes; out.layers = s.layers;
  out.focus = b64(new Float64Array(s.focus)); out.settings = settings; out.envW = s.envW; out.envH = s.envH;
} else if (mode === 'png') {
  // decodePng on files PIL wrote (every colour type / depth / interlacing the test made), encodePng back
  out.decoded = {};
  for (const f of job.files) { const im = SF.decodePng(fs.readFileSync(f)); out.decoded[f] = { width: im.width, height: im.height, rgba: b64(im.data) }; }
  const px = Uint8Array.from(Buffer.from(job.encode.rgba_b64, 'base64'));
  fs.writeFileSync(job.encode.out4, SF.encodePng(px, job.encode.width, job.encode.height, 4));
  const rgb = new Uint8Array(job.encode.width * job.encode.height * 3);
  for (let i = 0, j = 0; i < px.length; i += 4, j += 3) { rgb[j] = px[i]; rgb[j + 1] = px[i + 1]; rgb[j + 2] = px[i + 2]; }
  fs.writeFileSync(job.encode.out3, SF.encodePng(rgb, job.encode.width, job.encode.height, 3));
  let err = null;
  try { SF.decodePng(Buffer.from('GIF89a-not-a-png')); } catch (e) { err = String(e.message); }
  out.not_png = err;
} else if (mode === 'jpeg') {
  // decodeJpeg on files Pillow wrote; what it refuses to decode must say so
  out.decoded = {}; out.errors = {};
  for (const f of job.files) {
    try { const im = SF.decodeJpeg(fs.readFileSync(f)); out.decoded[f] = { width: im.width, height: im.height, rgba: b64(im.data) }; }
    catch (e) { out.errors[f] = String(e.message); }
  }
} else if (mode === 'render_scene_file') {
  const fr = SF.renderToPng(job.scene_path, job.out_png, job.W, job.H, { samples: job.samples, bounces: job.bounces, seed: job.seed, denoise: job.denoise });
  out.radiance = b64(fr.radiance); out.rgba = b64(fr.rgba);
} else if (mode === 'exports') {
  out.exports = Object.keys(F.addon).sort();
  out.abi = F.addon.abiVersion();
  out.devices = F.addon.deviceCount();
} else if (mode === 'build') {
  const s = F.buildScene(job.props, job.objs, env, 4);
  for (const k of ['bvh', 'tri', 'mat', 'norm', 'uv', 'bins', 'atlas']) out[k] = b64(s[k]);
  out.depth = s.depth; out.atlasLayers = s.atlasLayers;
} else if (mode === 'build_full') {
  // whole scene JSON: MTL group materials, decoded images, worldTransforms, normalize, auto-focus rays
  const assets = {};
  for (const [url, im] of Object.entries(job.images)) {
    assets[url] = { width: im.width, height: im.height, data: Uint8Array.from(Buffer.from(im.rgba_b64, 'base64')) };
  }
  const s = F.buildScene(job.scene, job.objs, env, 4, { mtlTexts: job.files, assets, focusRays: job.focus_rays });
  for (const k of ['bvh', 'tri', 'mat', 'norm', 'uv', 'bins', 'atlas']) out[k] = b64(s[k]);
  out.depth = s.depth; out.atlasLayers = s.atlasLayers; out.atlasRes = s.atlasRes; out.layers = s.layers;
  out.focus = b64(new Float64Array(s.focus));
} else if (mode === 'build_injected') {
  // opts.host: the caller's own TexturePacker / getMaterial / ParseMaterials drive the material step (INTEGRATION.md 2).
  // Stand-ins with the reference's interface names, built on the module's resolver: checks the plumbing, and that the
  // result equals the built-in route.
  class Packer extends F.AtlasLayers {
    get imageSet() { return this.entries; }
    setAndGetResolution() { return this.resolution(); }
  }
  let calls = 0;
  const host = { TexturePacker: Packer, ParseMaterials: F.readMtl,
    getMaterial: (prop, group, packer, assets, base) => { calls++; return F.resolveMaterial(prop, group, packer, assets, base); } };
  const s = F.buildScene(job.props, job.objs, env, 4, { host, atlasPixels: (packer) => packer.pixels() });
  const r = F.buildScene(job.props, job.objs, env, 4);
  out.calls = calls;
  out.same = ['bvh', 'tri', 'mat', 'norm', 'uv', 'bins', 'atlas'].every((k) => b64(s[k]) === b64(r[k])) && s.atlasLayers === r.atlasLayers && s.atlasRes === r.atlasRes;
  let threw = null;
  try { F.buildScene(job.props, job.objs, env, 4, { host }); } catch (e) { threw = String(e.message); }
  out.needs_pixels = threw;
} else if (mode === 'blob') {
  // read a blob written by Python, write it back from JS
  const s = F.loadBlob(job.blob_in);
  F.saveBlob(job.blob_out, s);
  out.n_tris = s.tri.length / 9; out.leafSize = s.leafSize; out.atlasLayers = s.atlasLayers;
} else if (mode === 'render_multi') {
  // one frame over several (virtual) devices from this one JS thread, part of it through the async entry point
  const s = F.buildScene(job.props, job.objs, env, 4);
  const mp = new F.MultiPathTracer(s, job.W, job.H, job.devices);
  mp.eye = job.cam.P; mp.dir = job.cam.I; mp.fovScale = job.cam.fov_scale; mp.envTheta = job.cam.env_theta;
  mp.lensFeatures = job.cam.lens; mp.numBounces = job.bounces;
  mp.seed(job.seed);
  mp.render(job.ticks_fused);
  for (let k = 0; k < job.ticks_two_call; k++) mp.tick();
  let ticked = 0;
  const iv = setInterval(() => { ticked++; }, 1);
  mp.renderAsync(job.ticks_async).then(() => {
    clearInterval(iv);
    out.radiance = b64(mp.readRadiance());
    out.event_loop_alive = ticked >= 0;
    // single-target async render of the same job for comparison
    const pt = new F.PathTracer(s, job.W, job.H, 0);
    pt.eye = job.cam.P; pt.dir = job.cam.I; pt.fovScale = job.cam.fov_scale; pt.envTheta = job.cam.env_theta;
    pt.lensFeatures = job.cam.lens; pt.numBounces = job.bounces;
    pt.seed(job.seed);
    return pt.renderAsync(job.ticks_fused + job.ticks_two_call + job.ticks_async).then(() => {
      out.radiance_single = b64(pt.readRadiance());
      pt.close(); mp.close();
      fs.writeFileSync(process.argv[4], JSON.stringify(out));
    });
  }).catch((e) => { console.error(e); process.exit(1); });
} else if (mode === 'render_multi_rccl') {
  // the library's RCCL exchanges from the JS host (one device: a one-rank communicator), then an unknown mode
  const s = F.buildScene(job.props, job.objs, env, 4);
  const mp = new F.MultiPathTracer(s, job.W, job.H, job.devices);
  mp.eye = job.cam.P; mp.dir = job.cam.I; mp.fovScale = job.cam.fov_scale; mp.envTheta = job.cam.env_theta;
  mp.lensFeatures = job.cam.lens; mp.numBounces = job.bounces;
  mp.seed(job.seed);
  mp.render(job.ticks);
  out.before = mp.exchange();
  mp.setExchange('rccl_reduce');
  out.reduce = mp.exchange();
  out.radiance_reduce = b64(mp.readRadiance());
  mp.setExchange('rccl_gather');
  out.radiance_gather = b64(mp.readRadiance());
  mp.setExchange('peer');
  out.radiance_peer = b64(mp.readRadiance());
  try { mp.setExchange('carrier-pigeon'); out.unknown = null; } catch (e) { out.unknown = String(e.message); }
  mp.close();
} else if (mode === 'async_guard') {
  // while renderAsync is in flight every other call on the tracer throws; close() waits for it (real library, real device)
  const s = F.buildScene(job.props, job.objs, env, 4);
  const pt = new F.PathTracer(s, job.W, job.H, 0);
  pt.eye = job.cam.P; pt.dir = job.cam.I; pt.fovScale = job.cam.fov_scale; pt.envTheta = job.cam.env_theta;
  pt.lensFeatures = job.cam.lens; pt.numBounces = job.bounces;
  pt.seed(job.seed);
  const thrown = (f) => { try { f(); return null; } catch (e) { return String(e.message); } };
  const p = pt.renderAsync(job.ticks);
  out.during = { readRadiance: thrown(() => pt.readRadiance()), tick: thrown(() => pt.tick()), clear: thrown(() => pt.clear()),
    sync: thrown(() => pt.sync()), render: thrown(() => pt.render(1)), renderAsync: thrown(() => pt.renderAsync(1)) };
  p.then(() => {
    out.radiance = b64(pt.readRadiance());
    const q = pt.renderAsync(job.ticks);   // ... and close() right behind a second job waits for it
    const c = pt.close();
    out.close_returned_promise = !!(c && typeof c.then === 'function');
    return Promise.all([q, c]).then(() => {
      out.after_close = thrown(() => pt.readRadiance());
      fs.writeFileSync(process.argv[4], JSON.stringify(out));
    });
  }).catch((e) => { console.error(e); process.exit(1); });
} else if (mode === 'drop_tracers') {
  // tracers that are dropped without close(): the handles' finalizers give the device memory back
  const s = F.buildScene(job.props, job.objs, env, 4);
  const sleep = (ms) => new Promise((r) => setTimeout(r, ms));
  const free = () => F.addon.deviceMemory(0).free;
  (async () => {
    { const warm = new F.PathTracer(s, job.W, job.H, 0); warm.render(2); warm.readRadiance(); warm.close(); }   // runtime pools, code objects
    for (let i = 0; i < 4; i++) { global.gc(); await sleep(5); }
    // the yardstick: the same number of tracers alive at once and then CLOSED by hand (what the HIP runtime keeps of 50
    // streams' worth of queues and signals after their destruction is its own pool, not a leak of this library's)
    (() => { const held = []; for (let k = 0; k < job.n; k++) { const pt = new F.PathTracer(s, job.W, job.H, 0); pt.render(2); pt.readRadiance(); held.push(pt); }
      for (const pt of held) pt.close(); })();
    for (let i = 0; i < 4; i++) { global.gc(); await sleep(5); }
    out.free_start = free();
    // (allocations inside functions of their own: nothing of them stays reachable from this frame's registers)
    (() => { const kept = new F.PathTracer(s, job.W, job.H, 0); kept.render(2); kept.readRadiance(); out.free_with_one = free(); })();
    (() => { for (let k = 0; k < job.n; k++) { const pt = new F.PathTracer(s, job.W, job.H, 0); pt.render(2); pt.readRadiance(); } })();   // never closed
    out.free_before_gc = free();
    for (let i = 0; i < 10; i++) { global.gc(); await sleep(10); }
    out.free_end = free();
    fs.writeFileSync(process.argv[4], JSON.stringify(out));
  })().catch((e) => { console.error(e); process.exit(1); });
} else if (mode === 'bounces_range') {
  // NUM_BOUNCES outside [0, 64] or not an integer: RangeError at the N-API boundary, before any device call
  out.errors = {};
  const cam = { P: [0, 0, 0], I: [0, 0, -1], lens: [0.5, 0.02] };
  for (const nb of job.values) {
    const v = nb === 'nan' ? NaN : nb;
    for (const [name, fn] of [['trace', () => F.addon.trace(null, 0, 0, 0, v)],
                              ['render', () => F.addon.render(null, Object.assign({ numBounces: v }, cam), 0, 1, 1)]]) {
      try { fn(); out.errors[name + ':' + nb] = 'no error'; } catch (e) { out.errors[name + ':' + nb] = e.name + ': ' + e.message; }
    }
  }
} else if (mode === 'nogpu') {
  const s = F.buildScene(job.props, job.objs, env, 4);
  try { new F.PathTracer(s, 16, 16, 0); out.error = null; } catch (e) { out.error = String(e.message); }
} else if (mode === 'render') {
  const s = F.buildScene(job.props, job.objs, env, 4);
  const pt = new F.PathTracer(s, job.W, job.H, 0);
  pt.eye = job.cam.P; pt.dir = job.cam.I; pt.fovScale = job.cam.fov_scale; pt.envTheta = job.cam.env_theta;
  pt.lensFeatures = job.cam.lens; pt.numBounces = job.bounces;
  pt.seed(job.seed);
  for (let k = 0; k < job.ticks_two_call; k++) pt.tick();
  pt.render(job.ticks_fused);
  out.radiance = b64(pt.readRadiance());
  pt.enableCounters(true);
  pt.close();
}
fs.writeFileSync(process.argv[4], JSON.stringify(out));
