// Build-container tooling: runs the REFERENCE's own JS scene pipeline
// (vector.js, bvh.js, obj_loader.js, env_sampler.js imported unmodified from a
// temp copy of /root/reference beside a {"type":"module"} package.json) and
// dumps what main.js would upload.  main.js itself cannot be imported (it
// touches the DOM at import time, main.js:953-975), so its pure array-pushing
// packing loops (main.js:360-392) and maskBVHBuffer (main.js:272-282) are
// restated here; getMaterial's result is supplied per prop by the caller.
import * as ObjLoader from './obj_loader.js';
import { BVH } from './bvh.js';
import { ProcessEnvRadiance } from './env_sampler.js';
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
  if (job.props) {
    const realLog = console.log;
    console.log = () => {};
    let geometry = [];
    for (const prop of job.props) {
      const parsed = await ObjLoader.parseMesh(job.objs[prop.path], prop, job.worldTransforms, 'x');
      Object.values(parsed.groups).forEach((group) => {
        group.triangles.forEach((t) => { t.material = prop.material; geometry.push(t); });
      });
    }
    const t0 = Date.now();
    const bvh = new BVH(geometry, job.leaf_size || 4);
    out.build_ms = Date.now() - t0;
    console.log = realLog;
    const bvhArray = bvh.serializeTree();
    let bvhBuffer = [], trianglesBuffer = [], materialBuffer = [], normalBuffer = [], uvBuffer = [];
    for (let i = 0; i < bvhArray.length; i++) {           // main.js:360-392
      let e = bvhArray[i];
      let node = e.node;
      let triIndex = node.leaf ? trianglesBuffer.length / 3 / 3 : -1;
      let bufferNode = [e.left, e.right, triIndex].concat(node.boundingBox.min, node.boundingBox.max);
      if (node.leaf) {
        let tris = node.getTriangles();
        for (let j = 0; j < tris.length; j++) {
          trianglesBuffer.push(...tris[j].verts[0], ...tris[j].verts[1], ...tris[j].verts[2]);
          let material = tris[j].material;
          materialBuffer.push(material.diffuseIndex, material.specularIndex, material.normalIndex,
            material.roughnessIndex, 0, 0, ...material.emittance, material.ior, material.dielectric, 0);
          for (let k = 0; k < 3; k++) {
            normalBuffer.push(...tris[j].normals[k], ...tris[j].tangents[k], ...tris[j].bitangents[k]);
          }
          uvBuffer.push(...tris[j].uvs[0], ...tris[j].uvs[1], ...tris[j].uvs[2]);
        }
      }
      for (let j = 0; j < bufferNode.length; j++) bvhBuffer.push(bufferNode[j]);
    }
    // maskBVHBuffer (main.js:272-282)
    let masked = new Float32Array(new Int32Array(bvhBuffer).buffer);
    for (let i = 0; i < bvhBuffer.length; i += 9) for (let j = 3; j < 9; j++) masked[i + j] = bvhBuffer[i + j];
    const b64 = (f32) => Buffer.from(f32.buffer, f32.byteOffset, f32.byteLength).toString('base64');
    out.depth = bvh.depth;
    out.bvh = b64(masked);
    out.tri = b64(new Float32Array(trianglesBuffer));
    out.mat = b64(new Float32Array(materialBuffer));
    out.norm = b64(new Float32Array(normalBuffer));
    out.uv = b64(new Float32Array(uvBuffer));
  }
  fs.writeFileSync(process.argv[3], JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
