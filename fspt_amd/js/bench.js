#!/usr/bin/env node
'use strict';
/* bench.js - bench.py's headline measurement from the reference's own host language (north_star: "host code stays
 * JavaScript (Node) calling the HIP kernels through a thin C-ABI N-API addon"): the scene comes off disk the way the
 * reference loads one (scene JSON -> OBJ / MTL / images -> initBVH: loadSceneFile = main.js:915-950 over buildScene), the
 * ticks go through PathTracer.render() (the fused form of main.js:838-857's tick loop), timed with process.hrtime around
 * exactly `steps` ticks + sync, `reps` regions, median.  Prints ONE JSON line with bench.py's keys.
 *
 *   node fspt_amd/js/bench.js --scene <web-root>/scene/bench.json [--width 1920 --height 1080 --bounces 8 --steps 20
 *        --warmup 5 --reps 7 --batch 128 --focal-depth 2 --aperture 0.02 --seed 1 --out-radiance frame.f32]
 * (tools/write_bench_scene.py writes bench.py's synthetic 'bunny' workload as such a web root.)                        */
const fs = require('fs');
const path = require('path');
const F = require('./fspt.js');
const SF = require('./scene_file.js');

function args(argv) {
  const o = { width: 1920, height: 1080, bounces: 8, steps: 20, warmup: 5, reps: 7, batch: 128, seed: 1, device: 0 };
  for (let i = 0; i < argv.length; i++) {
    const a = argv[i];
    if (!a.startsWith('--')) throw new Error('unexpected argument ' + a);
    const k = a.slice(2).replace(/-([a-z])/g, (_, c) => c.toUpperCase());
    const v = argv[++i];
    if (v === undefined) throw new Error(a + ' needs a value');
    o[k] = (k === 'scene' || k === 'outRadiance') ? v : Number(v);
  }
  if (!o.scene) throw new Error('--scene <scene.json> is required');
  for (const k of ['width', 'height', 'steps', 'reps']) if (!(o[k] >= 1)) throw new Error('--' + k + ' must be >= 1');
  return o;
}

function main() {
  const o = args(process.argv.slice(2));
  const t0 = process.hrtime.bigint();
  const { scene, settings } = SF.loadSceneFile(o.scene);
  const buildS = Number(process.hrtime.bigint() - t0) / 1e9;
  const pt = new F.PathTracer(scene, o.width, o.height, o.device);
  try {
    pt.eye = settings.eye.slice(); pt.dir = settings.dir.slice();
    pt.fovScale = settings.fovScale; pt.envTheta = settings.envTheta;
    // main.js:74 lensFeatures = [1 - 1 / focalDepth, apertureSize]; default: the auto-focus ray's distance (main.js:544)
    pt.lensFeatures = [o.focalDepth === undefined ? settings.focus : 1 - 1 / o.focalDepth, o.aperture === undefined ? settings.aperture : o.aperture];
    pt.numBounces = o.bounces;
    const batch = Math.max(1, Math.min(o.batch, Math.max(o.steps, o.warmup)));
    pt.setPipeline('wavefront', batch);
    pt.seed(o.seed);
    pt.setStageTiming(false);           // this host reads no per-kernel times: no event pairs around the launches
    pt.prepare();                       // path state is allocated here, never inside a timed region
    if (o.warmup > 0) pt.render(o.warmup);
    pt.sync();
    const ms = [];
    for (let r = 0; r < o.reps; r++) {
      const a = process.hrtime.bigint();
      pt.render(o.steps);
      pt.sync();
      ms.push(Number(process.hrtime.bigint() - a) / 1e6);
    }
    const sorted = ms.slice().sort((x, y) => x - y), med = sorted[(o.reps - 1) >> 1];
    const nTris = scene.tri.length / 9;
    const out = {
      metric: `Msamples/s at ${o.width}x${o.height} depth ${o.bounces} (${path.basename(o.scene, '.json')}, ${nTris} tri)`,
      value: Math.round(o.width * o.height * o.steps / (med / 1e3) / 1e6 * 1e3) / 1e3, unit: 'Msamples/s', n_gpus: 1,
      steps: o.steps, warmup: o.warmup, ms_per_step: Math.round(med / o.steps * 1e4) / 1e4, higher_is_better: true,
      scaling: 'weak', vs_baseline: null, dtype: 'f32', data: 'synthetic', reps: o.reps,
      rep_ms_per_step: ms.map((x) => Math.round(x / o.steps * 1e4) / 1e4),
      config: { workload: `${o.scene}: ${nTris} tri, ${o.width}x${o.height}, depth ${o.bounces}, 1 spp/step, lens [${pt.lensFeatures.map((x) => +x.toFixed(6))}]`,
        host: `node ${process.version} -> fspt_napi.node -> libfspt (C ABI)`, pipeline: 'wavefront', batch_ticks: batch,
        path_state_bytes: Number(pt.pathStateBytes().bytes), bvh_nodes: scene.bvh.length / 9, bvh_depth: scene.depth,
        env_bins: scene.bins.length / 4, scene_load_s: Math.round(buildS * 100) / 100,
        ticks_rendered: o.warmup + o.reps * o.steps, seed: o.seed },
    };
    if (o.outRadiance) {
      const rad = pt.readRadiance();
      fs.writeFileSync(o.outRadiance, Buffer.from(rad.buffer, rad.byteOffset, rad.byteLength));
    }
    console.log(JSON.stringify(out));
  } finally {
    pt.close();
  }
}

main();
