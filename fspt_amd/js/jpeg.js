'use strict';
/**
 * JPEG decoder (baseline, extended-sequential, progressive) for the Node host's scene loader (the reference hands .jpeg maps to the
 * browser's <img> decoder: asset_packs/dungeon/*.jpeg, texture_packer.js:103-121).
 *
 * A JPEG does not define its decoded bytes: they depend on the decoder's inverse DCT, chroma upsampling and colour
 * conversion.  This one reproduces what libjpeg / libjpeg-turbo do with their DEFAULT settings - the decoder behind
 * Pillow, i.e. behind the Python host (fspt_amd/scene_file.py), and behind Chromium - bit for bit:
 *   inverse DCT        the accurate integer one (JDCT_ISLOW, jidctint.c: 13-bit constants, two passes, PASS1_BITS 2)
 *   chroma upsampling  "fancy" triangle filters for 2:1 horizontal (h2v1) and 2:1 x 2:1 (h2v2), replication otherwise
 *   YCbCr -> RGB       jdcolor.c's 16-bit fixed-point tables
 * so that a scene with JPEG maps gives the same atlas bytes - and the same picture - from both hosts
 * (tests/test_node_host.py compares against Pillow on generated files: 4:4:4 / 4:2:2 / 4:2:0 / grey, optimised Huffman
 * tables, restart intervals, sizes that are not multiples of the MCU).
 * Baseline, extended-sequential and progressive Huffman files (a complete progressive file decodes to the same
 * coefficients: libjpeg's block smoothing only acts on files cut short).
 * Not decoded: lossless, arithmetic coding, 12-bit, CMYK / YCCK, 1:2 vertical-only subsampling.
 */

const ZIGZAG = new Uint8Array([
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]);

// jidctint.c: FIX(x) = round(x * 2^13)
const F_0_298 = 2446, F_0_390 = 3196, F_0_541 = 4433, F_0_765 = 6270, F_0_899 = 7373, F_1_175 = 9633, F_1_501 = 12299,
  F_1_847 = 15137, F_1_961 = 16069, F_2_053 = 16819, F_2_562 = 20995, F_3_072 = 25172;
const CONST_BITS = 13, PASS1_BITS = 2;

/** jpeg_idct_islow on one dequantised block (natural order, Int32Array(64)) -> 64 samples 0..255 into out[o + y * stride + x] */
function idctIslow(c, ws, out, o, stride) {
  // pass 1: columns -> ws (scaled up by 2^PASS1_BITS)
  for (let x = 0; x < 8; x++) {
    let z2 = c[16 + x], z3 = c[48 + x];
    let z1 = Math.imul(z2 + z3, F_0_541);
    let tmp2 = z1 + Math.imul(z3, -F_1_847), tmp3 = z1 + Math.imul(z2, F_0_765);
    z2 = c[x]; z3 = c[32 + x];
    let tmp0 = (z2 + z3) << CONST_BITS, tmp1 = (z2 - z3) << CONST_BITS;
    const tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = c[56 + x]; tmp1 = c[40 + x]; tmp2 = c[24 + x]; tmp3 = c[8 + x];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; let z4 = tmp1 + tmp3;
    const z5 = Math.imul(z3 + z4, F_1_175);
    tmp0 = Math.imul(tmp0, F_0_298); tmp1 = Math.imul(tmp1, F_2_053); tmp2 = Math.imul(tmp2, F_3_072); tmp3 = Math.imul(tmp3, F_1_501);
    z1 = Math.imul(z1, -F_0_899); z2 = Math.imul(z2, -F_2_562); z3 = Math.imul(z3, -F_1_961) + z5; z4 = Math.imul(z4, -F_0_390) + z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const r = 1 << (CONST_BITS - PASS1_BITS - 1), s = CONST_BITS - PASS1_BITS;
    ws[x] = (tmp10 + tmp3 + r) >> s; ws[56 + x] = (tmp10 - tmp3 + r) >> s;
    ws[8 + x] = (tmp11 + tmp2 + r) >> s; ws[48 + x] = (tmp11 - tmp2 + r) >> s;
    ws[16 + x] = (tmp12 + tmp1 + r) >> s; ws[40 + x] = (tmp12 - tmp1 + r) >> s;
    ws[24 + x] = (tmp13 + tmp0 + r) >> s; ws[32 + x] = (tmp13 - tmp0 + r) >> s;
  }
  // pass 2: rows -> samples (descale by 2^(CONST_BITS + PASS1_BITS + 3), + 128, clamp)
  for (let y = 0; y < 8; y++) {
    const w = y * 8;
    let z2 = ws[w + 2], z3 = ws[w + 6];
    let z1 = Math.imul(z2 + z3, F_0_541);
    let tmp2 = z1 + Math.imul(z3, -F_1_847), tmp3 = z1 + Math.imul(z2, F_0_765);
    let tmp0 = (ws[w] + ws[w + 4]) << CONST_BITS, tmp1 = (ws[w] - ws[w + 4]) << CONST_BITS;
    const tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = ws[w + 7]; tmp1 = ws[w + 5]; tmp2 = ws[w + 3]; tmp3 = ws[w + 1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; let z4 = tmp1 + tmp3;
    const z5 = Math.imul(z3 + z4, F_1_175);
    tmp0 = Math.imul(tmp0, F_0_298); tmp1 = Math.imul(tmp1, F_2_053); tmp2 = Math.imul(tmp2, F_3_072); tmp3 = Math.imul(tmp3, F_1_501);
    z1 = Math.imul(z1, -F_0_899); z2 = Math.imul(z2, -F_2_562); z3 = Math.imul(z3, -F_1_961) + z5; z4 = Math.imul(z4, -F_0_390) + z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    const s = CONST_BITS + PASS1_BITS + 3, r = 1 << (s - 1), p = o + y * stride;
    out[p] = limit((tmp10 + tmp3 + r) >> s); out[p + 7] = limit((tmp10 - tmp3 + r) >> s);
    out[p + 1] = limit((tmp11 + tmp2 + r) >> s); out[p + 6] = limit((tmp11 - tmp2 + r) >> s);
    out[p + 2] = limit((tmp12 + tmp1 + r) >> s); out[p + 5] = limit((tmp12 - tmp1 + r) >> s);
    out[p + 3] = limit((tmp13 + tmp0 + r) >> s); out[p + 4] = limit((tmp13 - tmp0 + r) >> s);
  }
}
// range_limit[x & RANGE_MASK] of jdmaster.c (prepare_range_limit_table): x + 128 clamped to 0..255 for -384 <= x < 512 + 384
// (the table's wrap-around halves make far-off values clamp the way they overflowed)
function limit(x) {
  x &= 1023;
  if (x < 128) return x + 128;      // 0 .. 127        -> 128 .. 255
  if (x < 512) return 255;          // 128 .. 511      -> 255
  if (x < 896) return 0;            // -512 .. -129    -> 0
  return x - 896;                   // -128 .. -1      -> 0 .. 127
}

function buildHuffman(counts, symbols) {
  // canonical codes: maxcode[l] / valptr[l] / mincode[l] as in jdhuff.c, plus an 9-bit look-ahead table
  const maxcode = new Int32Array(18).fill(-1), valptr = new Int32Array(17), mincode = new Int32Array(17);
  const look = new Int16Array(512).fill(-1); // (length << 8) | symbol
  let code = 0, k = 0;
  for (let l = 1; l <= 16; l++) {
    valptr[l] = k; mincode[l] = code;
    for (let i = 0; i < counts[l - 1]; i++, k++, code++) {
      if (l <= 9) {
        const lo = code << (9 - l), n = 1 << (9 - l);
        for (let j = 0; j < n; j++) look[lo + j] = (l << 8) | symbols[k];
      }
    }
    maxcode[l] = counts[l - 1] ? code - 1 : -1;
    code <<= 1;
  }
  maxcode[17] = 0x7fffffff;
  return { maxcode, valptr, mincode, look, symbols };
}

function decodeJpeg(buf) {
  if (!(buf.length > 4 && buf[0] === 0xFF && buf[1] === 0xD8)) throw new Error('not a JPEG file');
  const qt = [null, null, null, null], hdc = [null, null, null, null], hac = [null, null, null, null];
  let frame = null, restartInterval = 0, adobe = -1, pos = 2;
  const u16 = (p) => (buf[p] << 8) | buf[p + 1];
  let coefs = null; // per component: Int16Array of blocksW * blocksH * 64 (zig-zag order undone at store time)

  function setupFrame(p, len, marker) {
    if (marker !== 0xC0 && marker !== 0xC1 && marker !== 0xC2) {
      const what = (marker === 0xC3 || marker === 0xC7 || marker === 0xCB || marker === 0xCF) ? 'lossless'
        : marker >= 0xC9 ? 'arithmetic-coded' : 'hierarchical / differential';
      throw new Error(what + ' JPEG (SOF' + (marker - 0xC0) + ') is not decoded by the Node host: re-save it as a baseline JPEG or a PNG');
    }
    if (buf[p] !== 8) throw new Error(buf[p] + '-bit JPEG is not decoded (8-bit only)');
    const h = u16(p + 1), w = u16(p + 3), n = buf[p + 5];
    if (!w || !h) throw new Error('JPEG with zero size');
    if (w * h > (1 << 28)) throw new Error('JPEG of ' + w + ' x ' + h + ' pixels is not decoded (more than 2^28)');
    if (n !== 1 && n !== 3) throw new Error(n + '-component JPEG (CMYK / YCCK) is not decoded');
    if (len < 8 + 3 * n) throw new Error('truncated SOF segment');
    const comps = [];
    let hmax = 1, vmax = 1;
    for (let i = 0; i < n; i++) {
      const c = { id: buf[p + 6 + 3 * i], h: buf[p + 7 + 3 * i] >> 4, v: buf[p + 7 + 3 * i] & 15, tq: buf[p + 8 + 3 * i] };
      if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) throw new Error('bad JPEG component');
      hmax = Math.max(hmax, c.h); vmax = Math.max(vmax, c.v);
      comps.push(c);
    }
    const mcuW = 8 * hmax, mcuH = 8 * vmax, mcusX = Math.ceil(w / mcuW), mcusY = Math.ceil(h / mcuH);
    for (const c of comps) {
      c.bw = mcusX * c.h; c.bh = mcusY * c.v;                       // blocks incl. MCU padding (interleaved scans)
      c.dw = Math.ceil(w * c.h / hmax); c.dh = Math.ceil(h * c.v / vmax); // the component's true (downsampled) size
      c.coef = new Int16Array(c.bw * c.bh * 64);
      c.pred = 0;
    }
    frame = { w, h, comps, hmax, vmax, mcusX, mcusY, progressive: marker === 0xC2 };
  }

  function decodeScan(p, len) {
    const ns = buf[p];
    if (len < 6 + 2 * ns) throw new Error('truncated SOS segment');
    const sc = [];
    for (let i = 0; i < ns; i++) {
      const id = buf[p + 1 + 2 * i], c = frame.comps.find((x) => x.id === id);
      if (!c) throw new Error('SOS names an unknown component');
      c.td = buf[p + 2 + 2 * i] >> 4; c.ta = buf[p + 2 + 2 * i] & 15;
      sc.push(c);
    }
    const ss = buf[p + 1 + 2 * ns], se = buf[p + 2 + 2 * ns], ah = buf[p + 3 + 2 * ns] >> 4, al = buf[p + 3 + 2 * ns] & 15;
    const prog = frame.progressive;
    if (!prog && (ss !== 0 || se !== 63 || ah !== 0 || al !== 0)) throw new Error('not a sequential JPEG scan');
    if (prog && (ss > se || se > 63 || (ss === 0 && se !== 0) || (ss > 0 && ns !== 1) || al > 13)) throw new Error('bad progressive JPEG scan');
    for (const c of sc) if (((!prog || ss === 0) && !(prog && ah) && !hdc[c.td]) || ((!prog || ss > 0) && !hac[c.ta])) throw new Error('SOS uses an undefined Huffman table');
    let q = p + len - 2; // first entropy-coded byte
    // bit reader over the entropy-coded segment (0xFF00 -> 0xFF; stops feeding at a marker)
    let bits = 0, nbits = 0, hitMarker = false;
    const fill = () => {
      while (nbits <= 24) {
        let b = 0;
        if (!hitMarker && q < buf.length) {
          b = buf[q];
          if (b === 0xFF) {
            const b2 = q + 1 < buf.length ? buf[q + 1] : 0xD9;
            if (b2 === 0) q += 2;
            else { hitMarker = true; b = 0; }
          } else q++;
        }
        bits = (bits << 8) | b; nbits += 8; // (bits is used modulo 2^32: only the low nbits matter)
      }
    };
    const getBits = (n) => { if (nbits < n) fill(); nbits -= n; return (bits >>> nbits) & ((1 << n) - 1); };
    const decodeSym = (t) => {
      if (nbits < 16) fill();
      const e = t.look[(bits >>> (nbits - 9)) & 511];
      if (e >= 0) { nbits -= e >> 8; return e & 255; }
      let l = 10, code = (bits >>> (nbits - 10)) & 1023;
      while (code > t.maxcode[l]) { l++; if (l > 16) throw new Error('corrupt JPEG: bad Huffman code'); code = (bits >>> (nbits - l)) & ((1 << l) - 1); }
      nbits -= l;
      return t.symbols[t.valptr[l] + code - t.mincode[l]];
    };
    const extend = (v, s) => (v < (1 << (s - 1)) ? v - (1 << s) + 1 : v);
    const block = (c, bx, by) => {
      const o = (by * c.bw + bx) * 64, co = c.coef, dc = hdc[c.td], ac = hac[c.ta];
      const s = decodeSym(dc);
      if (s > 15) throw new Error('corrupt JPEG: bad DC size');
      c.pred += s ? extend(getBits(s), s) : 0;
      co[o] = c.pred;
      for (let k = 1; k < 64;) {
        const rs = decodeSym(ac), r = rs >> 4, sz = rs & 15;
        if (sz === 0) { if (r === 15) { k += 16; continue; } break; }
        k += r;
        if (k > 63) throw new Error('corrupt JPEG: run past the block');
        co[o + ZIGZAG[k]] = extend(getBits(sz), sz);
        k++;
      }
    };
    // progressive scans (jdphuff.c): DC first / refinement, AC first / refinement with end-of-band runs
    let eobrun = 0;
    const blockProg = (c, bx, by) => {
      const o = (by * c.bw + bx) * 64, co = c.coef;
      if (ss === 0) {
        if (ah === 0) {
          const s = decodeSym(hdc[c.td]);
          if (s > 15) throw new Error('corrupt JPEG: bad DC size');
          c.pred += s ? extend(getBits(s), s) : 0;
          co[o] = c.pred << al;
        } else if (getBits(1)) co[o] |= 1 << al;
        return;
      }
      const ac = hac[c.ta];
      if (ah === 0) {
        if (eobrun > 0) { eobrun--; return; }
        for (let k = ss; k <= se; k++) {
          const rs = decodeSym(ac), r = rs >> 4, sz = rs & 15;
          if (sz) {
            k += r;
            if (k > 63) throw new Error('corrupt JPEG: run past the block');
            co[o + ZIGZAG[k]] = extend(getBits(sz), sz) << al;
          } else if (r === 15) k += 15;
          else { eobrun = (1 << r) + (r ? getBits(r) : 0) - 1; break; }
        }
        return;
      }
      const p1 = 1 << al, m1 = -1 << al;
      const correct = (i) => { if (getBits(1) && (co[i] & p1) === 0) co[i] += co[i] >= 0 ? p1 : m1; };
      let k = ss;
      if (eobrun === 0) {
        for (; k <= se; k++) {
          const rs = decodeSym(ac);
          let r = rs >> 4, sv = rs & 15;
          if (sv) sv = getBits(1) ? p1 : m1;
          else if (r !== 15) { eobrun = (1 << r) + (r ? getBits(r) : 0); break; }
          do {
            const i = o + ZIGZAG[k];
            if (co[i] !== 0) correct(i);
            else if (--r < 0) break;
            k++;
          } while (k <= se);
          if (sv && k <= 63) co[o + ZIGZAG[k]] = sv;
        }
      }
      if (eobrun > 0) {
        for (; k <= se; k++) { const i = o + ZIGZAG[k]; if (co[i] !== 0) correct(i); }
        eobrun--;
      }
    };
    const restart = () => {
      // the next marker must be RSTn: drop the bit buffer, step over it
      nbits = 0; bits = 0;
      while (q < buf.length && !(buf[q] === 0xFF && buf[q + 1] >= 0xD0 && buf[q + 1] <= 0xD7)) q++;
      if (q < buf.length) q += 2;
      hitMarker = false;
      for (const c of sc) c.pred = 0;
      eobrun = 0;
    };
    for (const c of sc) c.pred = 0;
    const one = prog ? blockProg : block;
    if (ns === 1) {
      // non-interleaved: the component's own blocks (ceil(dw / 8) x ceil(dh / 8)), row-major
      const c = sc[0], nbx = Math.ceil(c.dw / 8), nby = Math.ceil(c.dh / 8);
      let n = 0;
      for (let by = 0; by < nby; by++)
        for (let bx = 0; bx < nbx; bx++) {
          if (restartInterval && n && n % restartInterval === 0) restart();
          one(c, bx, by); n++;
        }
    } else {
      let n = 0;
      for (let my = 0; my < frame.mcusY; my++)
        for (let mx = 0; mx < frame.mcusX; mx++) {
          if (restartInterval && n && n % restartInterval === 0) restart();
          for (const c of sc)
            for (let v = 0; v < c.v; v++)
              for (let h = 0; h < c.h; h++) one(c, mx * c.h + h, my * c.v + v);
          n++;
        }
    }
    // back to the marker that ended the scan
    if (!hitMarker) { while (q + 1 < buf.length && !(buf[q] === 0xFF && buf[q + 1] !== 0 && !(buf[q + 1] >= 0xD0 && buf[q + 1] <= 0xD7))) q++; }
    return q;
  }

  let sawEOI = false, scans = 0;
  while (pos + 4 <= buf.length && !sawEOI) {
    if (buf[pos] !== 0xFF) { pos++; continue; }
    const m = buf[pos + 1];
    if (m === 0xFF) { pos++; continue; }
    if (m === 0xD9) { sawEOI = true; break; }
    if (m === 0x01 || (m >= 0xD0 && m <= 0xD8)) { pos += 2; continue; }
    const len = u16(pos + 2), p = pos + 4;
    if (len < 2 || pos + 2 + len > buf.length) throw new Error('truncated JPEG segment');
    if (m === 0xDB) {
      for (let q = p; q < pos + 2 + len;) {
        const pq = buf[q] >> 4, tq = buf[q] & 15; q++;
        if (tq > 3) throw new Error('bad DQT');
        const t = new Int32Array(64);
        for (let k = 0; k < 64; k++) { t[ZIGZAG[k]] = pq ? u16(q) : buf[q]; q += pq ? 2 : 1; }
        qt[tq] = t;
      }
    } else if (m === 0xC4) {
      for (let q = p; q < pos + 2 + len;) {
        const tc = buf[q] >> 4, th = buf[q] & 15; q++;
        if (tc > 1 || th > 3) throw new Error('bad DHT');
        const counts = buf.slice(q, q + 16); q += 16;
        let n = 0; for (let i = 0; i < 16; i++) n += counts[i];
        if (n > 256 || q + n > pos + 2 + len) throw new Error('bad DHT');
        const t = buildHuffman(counts, buf.slice(q, q + n)); q += n;
        if (tc) hac[th] = t; else hdc[th] = t;
      }
    } else if (m >= 0xC0 && m <= 0xCF && m !== 0xC4 && m !== 0xC8 && m !== 0xCC) {
      if (frame) throw new Error('JPEG with more than one frame');
      setupFrame(p, len, m);
    } else if (m === 0xDD) {
      restartInterval = u16(p);
    } else if (m === 0xEE) {
      if (len >= 14 && buf.slice(p, p + 5).toString('latin1') === 'Adobe') adobe = buf[p + 11];
    } else if (m === 0xDA) {
      if (!frame) throw new Error('JPEG scan before the frame header');
      pos = decodeScan(p, len); scans++;
      continue;
    }
    pos += 2 + len;
  }
  if (!frame || !scans) throw new Error('JPEG without image data');

  // ---- dequantise + inverse DCT: every component at its own resolution (padded to whole blocks) ----
  const { w, h, comps, hmax, vmax } = frame;
  const ws = new Int32Array(64), blk = new Int32Array(64);
  for (const c of comps) {
    const q = qt[c.tq];
    if (!q) throw new Error('JPEG component without a quantisation table');
    c.stride = c.bw * 8;
    c.pix = new Uint8Array(c.stride * c.bh * 8);
    for (let by = 0; by < c.bh; by++)
      for (let bx = 0; bx < c.bw; bx++) {
        const o = (by * c.bw + bx) * 64;
        for (let k = 0; k < 64; k++) blk[k] = c.coef[o + k] * q[k];
        idctIslow(blk, ws, c.pix, by * 8 * c.stride + bx * 8, c.stride);
      }
    c.coef = null;
  }

  // ---- upsample the components to the frame's size ----
  const full = comps.map((c) => {
    if (c.h === hmax && c.v === vmax) return { pix: c.pix, stride: c.stride };
    const out = new Uint8Array(w * h);
    const hx = hmax / c.h, vy = vmax / c.v;
    const fancy = c.dw > 2; // jdsample.c: the triangle filters only for components more than two samples wide
    if (hx === 2 && vy === 1 && fancy) {
      // h2v1_fancy_upsample (jdsample.c): 3/4 nearer + 1/4 further, biases 1 (left output) and 2 (right output)
      for (let y = 0; y < h; y++) {
        const r = y * c.stride, n = c.dw, o = y * w;
        const put = (x, v) => { if (x < w) out[o + x] = v; };
        put(0, c.pix[r]); put(1, (c.pix[r] * 3 + c.pix[r + 1] + 2) >> 2);
        for (let i = 1; i < n - 1; i++) {
          const v3 = c.pix[r + i] * 3;
          put(2 * i, (v3 + c.pix[r + i - 1] + 1) >> 2); put(2 * i + 1, (v3 + c.pix[r + i + 1] + 2) >> 2);
        }
        put(2 * n - 2, (c.pix[r + n - 1] * 3 + c.pix[r + n - 2] + 1) >> 2); put(2 * n - 1, c.pix[r + n - 1]);
      }
    } else if (hx === 2 && vy === 2 && fancy) {
      // h2v2_fancy_upsample: vertically 3/4 nearer row + 1/4 further row (the rows beyond the image repeat its edge
      // rows), then horizontally the same with biases 8 / 7 on 16ths
      const n = c.dw, sum = new Int32Array(n);
      for (let y = 0; y < h; y++) {
        const v = y >> 1, v1 = Math.min(Math.max((y & 1) ? v + 1 : v - 1, 0), c.dh - 1);
        const r0 = v * c.stride, r1 = v1 * c.stride, o = y * w;
        for (let i = 0; i < n; i++) sum[i] = c.pix[r0 + i] * 3 + c.pix[r1 + i];
        const put = (x, val) => { if (x < w) out[o + x] = val; };
        put(0, (sum[0] * 4 + 8) >> 4); put(1, (sum[0] * 3 + sum[1] + 7) >> 4);
        for (let i = 1; i < n - 1; i++) { put(2 * i, (sum[i] * 3 + sum[i - 1] + 8) >> 4); put(2 * i + 1, (sum[i] * 3 + sum[i + 1] + 7) >> 4); }
        put(2 * n - 2, (sum[n - 1] * 3 + sum[n - 2] + 8) >> 4); put(2 * n - 1, (sum[n - 1] * 4 + 7) >> 4);
      }
    } else if (Number.isInteger(hx) && Number.isInteger(vy) && !(hx === 1 && vy === 2)) {
      // int_upsample / h2v1 / h2v2 box replication for the other integral ratios
      for (let y = 0; y < h; y++) { const r = Math.floor(y / vy) * c.stride, o = y * w; for (let x = 0; x < w; x++) out[o + x] = c.pix[r + Math.floor(x / hx)]; }
    } else {
      throw new Error('JPEG chroma sampling ' + c.h + 'x' + c.v + ' of ' + hmax + 'x' + vmax + ' is not decoded by the Node host');
    }
    return { pix: out, stride: w };
  });

  // ---- colour conversion (jdcolor.c) -> straight RGBA8 ----
  const data = new Uint8Array(w * h * 4);
  if (comps.length === 1) {
    for (let y = 0; y < h; y++) for (let x = 0; x < w; x++) { const v = full[0].pix[y * full[0].stride + x], o = (y * w + x) * 4; data[o] = data[o + 1] = data[o + 2] = v; data[o + 3] = 255; }
  } else if (adobe === 0 || (adobe < 0 && comps[0].id === 82 && comps[1].id === 71 && comps[2].id === 66)) { // Adobe transform 0 / ids 'R' 'G' 'B': the components ARE R, G, B
    for (let y = 0; y < h; y++) for (let x = 0; x < w; x++) { const o = (y * w + x) * 4; for (let k = 0; k < 3; k++) data[o + k] = full[k].pix[y * full[k].stride + x]; data[o + 3] = 255; }
  } else {
    const crR = new Int32Array(256), cbB = new Int32Array(256), crG = new Int32Array(256), cbG = new Int32Array(256);
    for (let i = 0, x = -128; i < 256; i++, x++) {
      crR[i] = (91881 * x + 32768) >> 16;   // FIX(1.40200)
      cbB[i] = (116130 * x + 32768) >> 16;  // FIX(1.77200)
      crG[i] = -46802 * x;                  // FIX(0.71414)
      cbG[i] = -22554 * x + 32768;          // FIX(0.34414) (+ ONE_HALF)
    }
    const clamp = (v) => (v < 0 ? 0 : v > 255 ? 255 : v);
    for (let y = 0; y < h; y++)
      for (let x = 0; x < w; x++) {
        const Y = full[0].pix[y * full[0].stride + x], cb = full[1].pix[y * full[1].stride + x], cr = full[2].pix[y * full[2].stride + x], o = (y * w + x) * 4;
        data[o] = clamp(Y + crR[cr]); data[o + 1] = clamp(Y + ((cbG[cb] + crG[cr]) >> 16)); data[o + 2] = clamp(Y + cbB[cb]); data[o + 3] = 255;
      }
  }
  return { width: w, height: h, data };
}

module.exports = { decodeJpeg };
