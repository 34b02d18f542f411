'use strict';
// tests/test_napi_handles.py: the addon's handle logic against tests/napi_mock/libfspt_mock.c (no GPU).
// usage: node --expose-gc napi_handles_check.js <dir with fspt_napi.node + libfspt.so (mock)> <out.json>
const path = require('path'), fs = require('fs');
const addon = require(path.join(process.argv[2], 'fspt_napi.node'));
const live = () => addon.deviceCount();   // the mock's hook: library objects alive
const params = { P: [0, 0, 2], I: [0, 0, -1], lens: [0.5, 0.02] };
const desc = { bvh: new Float32Array(9), tri: new Float32Array(9), mat: new Float32Array(12), norm: new Float32Array(27), uv: new Float32Array(6),
  atlas: new Uint8Array(4), atlasRes: 1, atlasLayers: 1, env: null, envW: 0, envH: 0, bins: new Uint32Array(4), leafSize: 4 };
const thrown = (f) => { try { f(); return null; } catch (e) { return e.constructor.name + ': ' + e.message; } };
const sleep = (ms) => new Promise((r) => setTimeout(r, ms));
async function collect() { for (let i = 0; i < 6; i++) { global.gc(); await sleep(5); } }

(async () => {
  const out = {};
  // ---- 1: every other call on a target throws while renderAsync is in flight; the scene cannot be destroyed either
  const scene = addon.sceneCreate(desc, 0), target = addon.targetCreate(scene, 2, 2);
  const buf = new Float32Array(16);
  const job = addon.renderAsync(target, params, 0, 1, 1);
  out.during = {
    readRadiance: thrown(() => addon.readRadiance(target, buf)),
    camera: thrown(() => addon.camera(target, [0, 0, 2], [0, 0, -1], 0.5, [0.5, 0.02], 1)),
    trace: thrown(() => addon.trace(target, 0, 1, 0, 4)),
    clear: thrown(() => addon.clear(target)),
    sync: thrown(() => addon.sync(target)),
    render: thrown(() => addon.render(target, params, 0, 1, 1)),
    renderAsync: thrown(() => addon.renderAsync(target, params, 0, 1, 1)),
    targetDestroy: thrown(() => addon.targetDestroy(target)),
    sceneDestroy: thrown(() => addon.sceneDestroy(scene)),
  };
  let first_after = null;
  await job.then(() => { first_after = thrown(() => addon.readRadiance(target, buf)); });   // the promise's first reaction may use the target
  out.first_reaction = first_after;
  out.renders_seen = buf[0];
  // ---- 2: kinds and destroyed handles
  out.kinds = {
    scene_as_target: thrown(() => addon.clear(scene)),
    target_as_scene: thrown(() => addon.targetCreate(target, 2, 2)),
    number_as_target: thrown(() => addon.clear(7)),
    scene_with_targets: thrown(() => addon.sceneDestroy(scene)),
  };
  addon.targetDestroy(target);
  out.kinds.destroyed_target = thrown(() => addon.clear(target));
  out.kinds.double_destroy = thrown(() => addon.targetDestroy(target));
  addon.sceneDestroy(scene);
  out.live_after_explicit_destroy = live();
  // ---- 3: dropped without destroy -> the finalizers give everything back, scene after its targets
  (() => { for (let i = 0; i < 50; i++) { const s = addon.sceneCreate(desc, 0); addon.targetCreate(s, 2, 2); addon.targetCreate(s, 2, 2); addon.builderCreate(); } })();
  out.live_before_gc = live();
  await collect();
  out.live_after_gc = live();
  // ---- 4: a job keeps its target alive even when JS drops every reference while it runs
  let p2 = (() => { const s = addon.sceneCreate(desc, 0), t = addon.targetCreate(s, 2, 2); return addon.renderAsync(t, params, 0, 1, 1); })();
  await collect();
  out.live_during_dropped_job = live();
  await p2; p2 = null;
  await collect();
  out.live_after_dropped_job = live();
  // ---- 5: a multi, its per-device target handle, and the job on the multi
  const multi = addon.multiCreate(desc, [0, 0], 2, 2), mt = addon.multiTarget(multi, 0);
  const mjob = addon.multiRenderAsync(multi, params, 0, 1, 1);
  out.multi_during = {
    multiReadRadiance: thrown(() => addon.multiReadRadiance(multi, buf)),
    its_target: thrown(() => addon.clear(mt)),
    multiDestroy: thrown(() => addon.multiDestroy(multi)),
  };
  await mjob;
  out.multi_after = { its_target: thrown(() => addon.clear(mt)), destroy_its_target: thrown(() => addon.targetDestroy(mt)),
    stage_ms: Array.from(addon.multiLastStageMs(multi, 2)) };
  addon.multiDestroy(multi);
  out.multi_after.its_target_after_destroy = thrown(() => addon.clear(mt));
  out.live_at_end = live();
  fs.writeFileSync(process.argv[3], JSON.stringify(out));
})().catch((e) => { console.error(e); process.exit(1); });
