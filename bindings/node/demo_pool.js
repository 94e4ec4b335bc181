// The hashing half of the SDK's worker pool on Node: batches of rows go to `concurrency` hashing workers round robin
// (pool.rs:84-104: worker = batch_idx % concurrency), the answers are put back in order by batch index
// (proving_worker.rs:154-159), and every digest is checked against Node's own BLAKE2s over the 32-byte-padded elements.
// usage: node demo_pool.js <libaero_stark.so> [rows] [width] [chunk] [concurrency]   -> one JSON line
'use strict';
const crypto = require('crypto');
const path = require('path');
const { Worker } = require('worker_threads');
const { encodeHashingWorkItem, decodeHashingResult } = require('./bincode');

const lib = path.resolve(process.argv[2]);
const rows = parseInt(process.argv[3] || '4096', 10), width = parseInt(process.argv[4] || '72', 10);
const chunk = parseInt(process.argv[5] || '1024', 10), concurrency = parseInt(process.argv[6] || '2', 10);
const P = 0xFFFFFFFF00000001n;

function hashElements(row) {                      // Blake2s_256::hash_elements: every element zero-padded to 32 bytes
  const b = Buffer.alloc(32 * row.length);
  row.forEach((e, i) => b.writeBigUInt64LE(e, 32 * i));
  return crypto.createHash('blake2s256').update(b).digest();
}
let seed = 0x9E3779B97F4A7C15n;
function next() {                                 // xorshift64*: a deterministic table of field elements
  seed ^= seed >> 12n; seed ^= (seed << 25n) & 0xFFFFFFFFFFFFFFFFn; seed ^= seed >> 27n;
  return ((seed * 0x2545F4914F6CDD1Dn) & 0xFFFFFFFFFFFFFFFFn) % P;
}
const table = [];
for (let r = 0; r < rows; r++) { const row = []; for (let c = 0; c < width; c++) row.push(next()); table.push(row); }

const workers = [];
for (let i = 0; i < concurrency; i++) workers.push(new Worker(path.join(__dirname, 'hashing_worker.js'), { workerData: { lib, device: 0 } }));
const nBatches = Math.ceil(rows / chunk);
const results = new Array(nBatches);
let pending = nBatches;
const t0 = process.hrtime.bigint();
for (const w of workers) {
  w.on('error', (e) => { console.log(JSON.stringify({ ok: false, error: String(e) })); process.exit(1); });
  w.on('message', (msg) => {
    if (!(msg instanceof Uint8Array)) return;
    const { batchIdx, hashes } = decodeHashingResult(msg);
    results[Number(batchIdx)] = hashes;
    if (--pending === 0) finish();
  });
}
for (let b = 0; b < nBatches; b++) workers[b % concurrency].postMessage(encodeHashingWorkItem(table.slice(b * chunk, (b + 1) * chunk), b));

function finish() {
  const ms = Number(process.hrtime.bigint() - t0) / 1e6;
  let bad = 0, r = 0;
  for (const hashes of results) for (const h of hashes) { if (!h.equals(hashElements(table[r]))) bad++; r++; }
  const ok = bad === 0 && r === rows;
  let alive = workers.length;
  for (const w of workers) {
    w.on('exit', () => { if (--alive === 0) process.exit(ok ? 0 : 1); });
    w.postMessage('close');
  }
  console.log(JSON.stringify({ ok, rows: r, width, batches: nBatches, concurrency, mismatches: bad, ms }));
}
