'use strict';
/**
 * scene_file.js — scene files, assets and frames for the JavaScript host (Node, CommonJS).
 *
 * What the reference does in PathTracer() / start() / tick() when it is pointed at `scene/<name>.json`
 * (main.js:915-950 collect + load the assets, utility.js:1-33 loadAll, main.js:284-445 initBVH,
 * main.js:838-866 tick / uploadOutput), with the browser's pieces replaced by Node's own:
 *   XHR text loads            -> fs.readFileSync
 *   <img> decoding            -> decodePng below (zlib.inflateSync + the five PNG filters): straight-alpha RGBA8,
 *                                row 0 = top - what a browser hands to texImage2D; baseline JPEG -> jpeg.js (libjpeg's
 *                                default decode, bit for bit: the bytes Pillow gives the Python host)
 *   canvas.toBlob('image/png')-> encodePng
 * The path tracing itself happens in libfspt through fspt.js.  Same semantics as fspt_amd/scene_file.py.
 */
const fs = require('fs');
const path = require('path');
const zlib = require('zlib');
const F = require('./fspt.js');
const { decodeJpeg } = require('./jpeg.js');

const PNG_SIG = Buffer.from([137, 80, 78, 71, 13, 10, 26, 10]);

/** PNG (all colour types, bit depths 1-16, non-interlaced and Adam7) -> {width, height, data: Uint8Array RGBA8}.
 *  16-bit samples keep their high byte; tRNS gives palette / colour-key transparency. */
function decodePng(buf) {
  if (buf.length < 8 || !buf.slice(0, 8).equals(PNG_SIG)) throw new Error('not a PNG file');
  let off = 8, ihdr = null, plte = null, trns = null;
  const idat = [];
  while (off + 8 <= buf.length) {
    const len = buf.readUInt32BE(off), type = buf.toString('latin1', off + 4, off + 8);
    const body = buf.slice(off + 8, off + 8 + len);
    if (type === 'IHDR') ihdr = body;
    else if (type === 'PLTE') plte = body;
    else if (type === 'tRNS') trns = body;
    else if (type === 'IDAT') idat.push(body);
    else if (type === 'IEND') break;
    off += 12 + len;
  }
  if (!ihdr || !idat.length) throw new Error('PNG without IHDR / IDAT');
  const width = ihdr.readUInt32BE(0), height = ihdr.readUInt32BE(4), depth = ihdr[8], ctype = ihdr[9], interlace = ihdr[12];
  const channels = { 0: 1, 2: 3, 3: 1, 4: 2, 6: 4 }[ctype];
  if (!channels || ![1, 2, 4, 8, 16].includes(depth)) throw new Error('unsupported PNG colour type ' + ctype + ' / depth ' + depth);
  if (ctype === 3 && !plte) throw new Error('palette PNG without PLTE');
  const raw = zlib.inflateSync(Buffer.concat(idat));
  const bpp = Math.max(1, (channels * depth) >> 3);           // bytes per complete pixel, for the filters
  const out = new Uint8Array(width * height * 4);
  const maxv = (1 << depth) - 1;
  const key = (trns && ctype === 0) ? trns.readUInt16BE(0) : null;
  const keyRgb = (trns && ctype === 2) ? [trns.readUInt16BE(0), trns.readUInt16BE(2), trns.readUInt16BE(4)] : null;

  function sample(line, i) {                                    // i-th sample of an unfiltered scanline
    if (depth === 8) return line[i];
    if (depth === 16) return (line[2 * i] << 8) | line[2 * i + 1];
    const per = 8 / depth, byte = line[Math.floor(i / per)];
    return (byte >> ((per - 1 - (i % per)) * depth)) & maxv;
  }
  const to8 = (v) => (depth === 8 ? v : depth === 16 ? v >> 8 : Math.round(v * 255 / maxv));

  function putPixel(line, x, ox, oy) {
    const o = (oy * width + ox) * 4;
    if (ctype === 3) {
      const i = sample(line, x);
      out[o] = plte[3 * i]; out[o + 1] = plte[3 * i + 1]; out[o + 2] = plte[3 * i + 2];
      out[o + 3] = (trns && i < trns.length) ? trns[i] : 255;
    } else if (ctype === 0 || ctype === 4) {
      const g = sample(line, x * channels);
      out[o] = out[o + 1] = out[o + 2] = to8(g);
      out[o + 3] = ctype === 4 ? to8(sample(line, x * 2 + 1)) : (key !== null && g === key ? 0 : 255);
    } else {
      const r = sample(line, x * channels), g = sample(line, x * channels + 1), b = sample(line, x * channels + 2);
      out[o] = to8(r); out[o + 1] = to8(g); out[o + 2] = to8(b);
      out[o + 3] = ctype === 6 ? to8(sample(line, x * 4 + 3)) : (keyRgb && r === keyRgb[0] && g === keyRgb[1] && b === keyRgb[2] ? 0 : 255);
    }
  }

  let pos = 0;
  function pass(pw, ph, place) {                                // one (sub)image of pw x ph pixels
    if (pw === 0 || ph === 0) return;
    const stride = (pw * channels * depth + 7) >> 3;
    let prev = new Uint8Array(stride);
    for (let y = 0; y < ph; y++) {
      const ft = raw[pos++];
      const line = new Uint8Array(raw.buffer, raw.byteOffset + pos, stride).slice();
      pos += stride;
      if (pos > raw.length) throw new Error('PNG data truncated');
      for (let i = 0; i < stride; i++) {
        const a = i >= bpp ? line[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
        let p = 0;
        if (ft === 1) p = a;
        else if (ft === 2) p = b;
        else if (ft === 3) p = (a + b) >> 1;
        else if (ft === 4) { const q = a + b - c, pa = Math.abs(q - a), pb = Math.abs(q - b), pc = Math.abs(q - c); p = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
        else if (ft !== 0) throw new Error('bad PNG filter type ' + ft);
        line[i] = (line[i] + p) & 255;
      }
      for (let x = 0; x < pw; x++) place(line, x, y);
      prev = line;
    }
  }
  if (!interlace) pass(width, height, (line, x, y) => putPixel(line, x, x, y));
  else {
    const A7 = [[0, 0, 8, 8], [4, 0, 8, 8], [0, 4, 4, 8], [2, 0, 4, 4], [0, 2, 2, 4], [1, 0, 2, 2], [0, 1, 1, 2]];
    for (const [x0, y0, dx, dy] of A7)
      pass(Math.ceil((width - x0) / dx), Math.ceil((height - y0) / dy), (line, x, y) => putPixel(line, x, x0 + x * dx, y0 + y * dy));
  }
  return { width, height, data: out };
}

const CRC = (() => { const t = new Uint32Array(256); for (let n = 0; n < 256; n++) { let c = n; for (let k = 0; k < 8; k++) c = (c & 1) ? (0xEDB88320 ^ (c >>> 1)) : (c >>> 1); t[n] = c >>> 0; } return t; })();
function crc32(buf) { let c = 0xFFFFFFFF; for (let i = 0; i < buf.length; i++) c = CRC[(c ^ buf[i]) & 255] ^ (c >>> 8); return (c ^ 0xFFFFFFFF) >>> 0; }
function chunk(type, body) {
  const head = Buffer.alloc(8); head.writeUInt32BE(body.length, 0); head.write(type, 4, 'latin1');
  const crc = Buffer.alloc(4); crc.writeUInt32BE(crc32(Buffer.concat([head.slice(4), body])), 0);
  return Buffer.concat([head, body, crc]);
}
/** rows top-first, `channels` = 3 (RGB) or 4 (RGBA) bytes per pixel -> PNG file bytes (8-bit, filter 0) */
function encodePng(pixels, width, height, channels) {
  const ch = channels || 4, stride = width * ch;
  if (pixels.length !== stride * height) throw new RangeError('encodePng: need width*height*channels bytes');
  const raw = Buffer.alloc((stride + 1) * height);
  for (let y = 0; y < height; y++) Buffer.from(pixels.buffer, pixels.byteOffset + y * stride, stride).copy(raw, y * (stride + 1) + 1);
  const ihdr = Buffer.alloc(13); ihdr.writeUInt32BE(width, 0); ihdr.writeUInt32BE(height, 4); ihdr[8] = 8; ihdr[9] = ch === 3 ? 2 : 6;
  return Buffer.concat([PNG_SIG, chunk('IHDR', ihdr), chunk('IDAT', zlib.deflateSync(raw)), chunk('IEND', Buffer.alloc(0))]);
}

function readImage(root, rel) {
  const file = path.join(root, rel);
  const buf = fs.readFileSync(file);
  if (buf.length >= 8 && buf.slice(0, 8).equals(PNG_SIG)) return Object.assign(decodePng(buf), { currentSrc: rel });
  if (buf.length >= 3 && buf[0] === 0xFF && buf[1] === 0xD8 && buf[2] === 0xFF) return Object.assign(decodeJpeg(buf), { currentSrc: rel });
  throw new Error(file + ': only PNG and JPEG (baseline / progressive) images are decoded by the Node host');
}

/** the urls obj_loader.js:185-187 fetches while parsing: basePath + '/' + the rest of each `mtllib` line */
function mtllibUrls(objText, basePath) {
  const urls = [];
  for (const line of objText.split('\n')) {
    const tok = line.trim().split(/[ ]+/);
    if (tok[0] === 'mtllib') urls.push(basePath + '/' + tok.slice(1).join(' '));
  }
  return urls;
}

/** PathTracer(scenePath ...) up to start() (main.js:915-950): read the scene JSON, collect and load every asset it
 *  names - OBJ texts, the MTL libraries of their `mtllib` lines, the texture images of prop fields and MTL `map_*`
 *  statements, the RGBE environment image - then initBVH through buildScene and the auto-focus ray.
 *  Returns {scene (buildScene's arrays), settings (initGlobals' values, main.js:50-75)}.  Paths in the JSON are relative
 *  to the web root (`assetRoot`, default: the parent of the scene file's folder). */
function loadSceneFile(scenePath, assetRoot, leafSize) {
  const sceneJson = JSON.parse(fs.readFileSync(scenePath, 'utf8'));
  const root = assetRoot || path.dirname(path.dirname(path.resolve(scenePath)));
  const props = F.mergeSceneProps(sceneJson);
  const objTexts = {}, mtlTexts = {}, assets = {};
  const wantImage = (url) => { if (!assets[url]) assets[url] = readImage(root, url); };
  for (const p of props) {
    if (objTexts[p.path] === undefined) objTexts[p.path] = fs.readFileSync(path.join(root, p.path), 'utf8');
    const base = p.path.split('/').slice(0, -1).join('/');
    for (const url of mtllibUrls(objTexts[p.path], base)) {
      if (mtlTexts[url] === undefined) mtlTexts[url] = fs.readFileSync(path.join(root, url), 'utf8');
      for (const tex of F.readMtl(mtlTexts[url], base).urls) wantImage(tex);
    }
    for (const k of ['diffuse', 'metallicRoughness']) if (typeof p[k] === 'string') wantImage(p[k]);
    for (const k of ['normal', 'emission']) if (p[k] && typeof p[k] === 'string') wantImage(p[k]);
  }
  let env = null;
  const e = sceneJson.environment;
  if (typeof e === 'string') { const img = readImage(root, e); env = { rgbe: img.data, width: img.width, height: img.height }; }
  else if (e && e.length) throw new Error('array-of-stops environments are not supported (broken in the reference itself); use an RGBE image or none');
  const eye = (sceneJson.cameraPos || [0, 0, 2]).map(Number), dir = (sceneJson.cameraDir || [0, 0, -1]).map(Number);
  const scene = F.buildScene(sceneJson, objTexts, env, leafSize || 4, { mtlTexts, assets, focusRays: [[eye, dir]] });
  const settings = { eye, dir, fovScale: Number(sceneJson.fovScale || 0.5), envTheta: Number(sceneJson.environmentTheta || 0),
    exposure: Number(sceneJson.exposure || 1.0), samples: Math.floor(Number(sceneJson.samples || 2000)), focus: scene.focus[0], aperture: 0.02 };
  return { scene, settings };
}

/** One frame as the reference produces it in frame mode (main.js:838-866): `samples` ticks from a cleared accumulator,
 *  then drawQuad.  Returns {rgba: Uint8Array (rows top-first, what canvas.toBlob encodes), radiance: Float32Array
 *  (rows bottom-first), width, height}. */
function renderFrame(scene, settings, width, height, opts) {
  opts = opts || {};
  const pt = new F.PathTracer(scene, width, height, opts.device || 0);
  try {
    pt.eye = settings.eye.slice(); pt.dir = settings.dir.slice();
    pt.fovScale = settings.fovScale; pt.envTheta = settings.envTheta;
    pt.lensFeatures = [settings.focus, settings.aperture];
    pt.numBounces = opts.bounces === undefined ? 4 : opts.bounces;
    pt.seed(opts.seed === undefined ? 1 : opts.seed);
    pt.render(opts.samples === undefined ? settings.samples : opts.samples);
    const canvas = pt.drawQuad(settings.exposure, opts.saturation, opts.denoise, opts.maxSigma);
    const radiance = pt.readRadiance();
    const rgba = new Uint8Array(canvas.length);
    for (let y = 0; y < height; y++) rgba.set(canvas.subarray((height - 1 - y) * width * 4, (height - y) * width * 4), y * width * 4);
    return { rgba, radiance, width, height };
  } finally {
    pt.close();
  }
}

/** scene file -> PNG file (the reference POSTs canvas.toBlob's PNG to /upload/<scene>/<frame>, main.js:859-866) */
function renderToPng(scenePath, outPath, width, height, opts) {
  const { scene, settings } = loadSceneFile(scenePath, opts && opts.assetRoot);
  const fr = renderFrame(scene, settings, width, height, opts);
  const rgb = new Uint8Array(width * height * 3);
  for (let i = 0, j = 0; i < fr.rgba.length; i += 4, j += 3) { rgb[j] = fr.rgba[i]; rgb[j + 1] = fr.rgba[i + 1]; rgb[j + 2] = fr.rgba[i + 2]; }
  fs.mkdirSync(path.dirname(path.resolve(outPath)), { recursive: true });
  fs.writeFileSync(outPath, encodePng(rgb, width, height, 3));
  return fr;
}

/** frame=N sequencing (main.js:851-866, 966-969): for every N of `frames` load scenePattern with {frame} replaced (the
 *  per-frame scene JSON the reference's server hands out for `?frame=N`), render it, write outPattern with {frame}
 *  replaced (the reference POSTs the canvas PNG to /upload/<scene>/<N>), go on to N + 1.  Returns the files written. */
function renderSequence(scenePattern, frames, outPattern, width, height, opts) {
  const written = [];
  for (const n of frames) {
    const out = outPattern.replace('{frame}', String(n));
    renderToPng(scenePattern.replace('{frame}', String(n)), out, width, height, opts);
    written.push(out);
  }
  return written;
}

module.exports = { decodePng, encodePng, decodeJpeg, mtllibUrls, loadSceneFile, renderFrame, renderToPng, renderSequence };

// node fspt_amd/js/scene_file.js scene/bunny.json out.png [--width W] [--height H] [--samples N] [--bounces B] [--seed S]
//                                [--asset-root DIR] [--denoise] [--frames A:B]   (mirrors `python -m fspt_amd.render`;
//                                with --frames both paths contain {frame}: the reference's ?frame=N loop)
if (require.main === module) {
  const argv = process.argv.slice(2), pos = [], o = { width: 960, height: 540 };
  for (let i = 0; i < argv.length; i++) {
    const a = argv[i];
    if (a === '--denoise') o.denoise = true;
    else if (a.startsWith('--')) o[a.slice(2).replace(/-([a-z])/g, (m, c) => c.toUpperCase())] = argv[++i];
    else pos.push(a);
  }
  if (pos.length !== 2) { console.error('usage: node scene_file.js <scene.json> <out.png> [--width W] [--height H] [--samples N] [--bounces B] [--seed S] [--asset-root DIR] [--denoise]'); process.exit(2); }
  const num = (k) => (o[k] === undefined ? undefined : Number(o[k]));
  const t0 = Date.now();
  const ro = { samples: num('samples'), bounces: num('bounces'), seed: num('seed'), assetRoot: o.assetRoot, denoise: !!o.denoise };
  if (o.frames) {
    const [a, b] = String(o.frames).split(':').map(Number), frames = [];
    for (let n = a; n < b; n++) frames.push(n);
    const written = renderSequence(pos[0], frames, pos[1], Number(o.width), Number(o.height), ro);
    console.log(JSON.stringify({ out: written, seconds: (Date.now() - t0) / 1000 }));
  } else {
    const fr = renderToPng(pos[0], pos[1], Number(o.width), Number(o.height), ro);
    console.log(JSON.stringify({ out: pos[1], width: fr.width, height: fr.height, seconds: (Date.now() - t0) / 1000 }));
  }
}
