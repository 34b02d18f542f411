// Build-container tooling: runs the REFERENCE's own JS scene pipeline
// (vector.js, bvh.js, obj_loader.js, env_sampler.js imported unmodified from a
// temp copy of /root/reference beside a {"type":"module"} package.json) and
// dumps what main.js would upload.  main.js itself cannot be imported (it
// touches the DOM at import time, main.js:953-975): js_ref.py cuts the source text of getMaterial (main.js:206-270),
// mergeSceneProps (:869-871), maskBVHBuffer (:272-282), shootAutoFocusRay (:447-546), the scene.normalize block
// (:337-348) and the packing loops (:358-392) out of main.js at run time into ref_functions.js (never committed), and
// they run unmodified against the reference's own modules.  Nothing of main.js is restated here.
import * as ObjLoader from './obj_loader.js';
import { BVH } from './bvh.js';
import { ProcessEnvRadiance } from './env_sampler.js';
import { TexturePacker } from './texture_packer.js';
import { BoundingBox } from './bvh.js';
import { Vec3 } from './vector.js';
import { getMaterial, mergeSceneProps, autoFocus, normalizeScene, packScene, maskBVHBuffer } from './ref_functions.js';
import fs from 'fs';

(async () => {
  const job = JSON.parse(fs.readFileSync(process.argv[2], 'utf8'));
  const out = {};
  if (job.env) {
    // DOM shim for env_sampler.js:49-55: the canvas 2D round trip is replaced by the exact bytes
    const data = Uint8Array.from(Buffer.from(job.env.rgba_b64, 'base64'));
    global.document = { createElement: () => ({ getContext: () => ({ drawImage() {}, getImageData: () => ({ data }) }) }) };
    const bins = ProcessEnvRadiance({ width: job.env.width, height: job.env.height });
    out.bins = Array.from(bins);
  }
  if (job.props || job.scene) {
    const realLog = console.log;
    console.log = () => {};
    let geometry = [];
    if (job.scene) {
      // initBVH's prop loop (main.js:311-335) with the reference's own getMaterial / TexturePacker / mergeSceneProps.
      // utility.js getText uses XMLHttpRequest: serve the job's files (mtllib texts).
      global.XMLHttpRequest = class {
        addEventListener(ev, fn) { if (ev === 'load') this.onload = fn; }
        open(method, url) { this.url = url; }
        send() {
          if (!(this.url in job.files)) throw new Error('no such file in job: ' + this.url);
          setTimeout(() => this.onload({ target: { responseText: job.files[this.url] } }), 0);
        }
      };
      const scene = job.scene;
      const props = mergeSceneProps(scene);
      const texturePacker = new TexturePacker(scene.atlasRes || 2048, props.length);
      // stand-ins for decoded HTMLImageElements: addTexture only looks at currentSrc and height
      const assets = {};
      for (const url of Object.keys(job.images || {})) assets[url] = { currentSrc: url, height: job.images[url].height };
      const bounds = new BoundingBox();
      for (let i = 0; i < props.length; i++) {
        const prop = props[i];
        const basePath = prop.path.split('/').slice(0, -1).join('/');
        const parsed = await ObjLoader.parseMesh(job.objs[prop.path], prop, scene.worldTransforms, basePath);
        bounds.addVertex(parsed.bounds.max);
        bounds.addVertex(parsed.bounds.min);
        Object.values(parsed.groups).forEach((group) => {
          const material = getMaterial(prop, group, texturePacker, assets, basePath);
          group.triangles.forEach((t) => { t.material = material; geometry.push(t); });
        });
      }
      normalizeScene(Vec3, scene, bounds, geometry);          // main.js:337-348, cut from main.js
      out.image_set = texturePacker.imageSet.map((e) => Array.isArray(e) ? { color: e } :
        { src: e.currentSrc, corrected: !!e.corrected, swizzle: e.swizzle || null });
      out.n_props = props.length;
    } else {
      for (const prop of job.props) {
        const parsed = await ObjLoader.parseMesh(job.objs[prop.path], prop, job.worldTransforms, 'x');
        Object.values(parsed.groups).forEach((group) => {
          group.triangles.forEach((t) => { t.material = prop.material; geometry.push(t); });
        });
      }
    }
    const t0 = Date.now();
    const bvh = new BVH(geometry, job.leaf_size || 4);
    out.build_ms = Date.now() - t0;
    // shootAutoFocusRay (main.js:447-546) needs the un-serialized tree: run it first
    out.autofocus = (job.autofocus || []).map(([eye, dir]) => autoFocus(Vec3, bvh, eye, dir));
    const { bvhBuffer, trianglesBuffer, materialBuffer, normalBuffer, uvBuffer } = packScene(bvh);   // main.js:358-392, cut from main.js
    const masked = maskBVHBuffer(bvhBuffer);                                                        // main.js:272-282
    console.log = realLog;
    // large scenes (job.raw_dir): the arrays go to binary files, the JSON carries only their names (a 1 M-triangle
    // scene's normals alone are 108 MB: as base64 inside one JSON string they pass V8's string limits)
    const b64 = (f32, name) => {
      const buf = Buffer.from(f32.buffer, f32.byteOffset, f32.byteLength);
      if (!job.raw_dir) return buf.toString('base64');
      fs.writeFileSync(job.raw_dir + '/' + name + '.f32', buf);
      return { file: name + '.f32' };
    };
    out.depth = bvh.depth;
    out.bvh = b64(masked, 'bvh');
    out.tri = b64(new Float32Array(trianglesBuffer), 'tri');
    out.mat = b64(new Float32Array(materialBuffer), 'mat');
    out.norm = b64(new Float32Array(normalBuffer), 'norm');
    out.uv = b64(new Float32Array(uvBuffer), 'uv');
  }
  fs.writeFileSync(process.argv[3], JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
