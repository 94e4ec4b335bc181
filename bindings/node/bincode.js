// bincode 1.3 messages of the SDK's worker seam (aero-sdk/miden-wasm/src/utils.rs:302-450), JavaScript side.
// Little endian, u64 lengths, usize as u64; a field element travels as its canonical u64 (BigInt here).
'use strict';

function u64(v) {
  const b = Buffer.alloc(8);
  b.writeBigUInt64LE(BigInt(v));
  return b;
}
function felts(values) {
  const b = Buffer.alloc(8 + 8 * values.length);
  b.writeBigUInt64LE(BigInt(values.length), 0);
  for (let i = 0; i < values.length; i++) b.writeBigUInt64LE(BigInt(values[i]), 8 + 8 * i);
  return b;
}
// HashingWorkItem { data: Vec<Vec<Felt>>, batch_idx: usize }  (utils.rs:358-362)
function encodeHashingWorkItem(rows, batchIdx) {
  return Buffer.concat([u64(rows.length), ...rows.map(felts), u64(batchIdx)]);
}
// HashingResult { batch_idx: usize, hashes: Vec<[u8; 32]> }  (utils.rs:411-415)
function decodeHashingResult(buf) {
  const b = Buffer.from(buf.buffer, buf.byteOffset, buf.length);
  const batchIdx = b.readBigUInt64LE(0);
  const n = Number(b.readBigUInt64LE(8));
  if (b.length !== 16 + 32 * n) throw new Error('HashingResult: unexpected length');
  const hashes = [];
  for (let i = 0; i < n; i++) hashes.push(b.slice(16 + 32 * i, 48 + 32 * i));
  return { batchIdx, hashes };
}
// ProverOutput { proof, program_outputs, public_inputs: Vec<u8> }  (utils.rs:424-430)
function decodeProverOutput(buf) {
  const b = Buffer.from(buf.buffer, buf.byteOffset, buf.length);
  const parts = [];
  let o = 0;
  for (let i = 0; i < 3; i++) {
    const n = Number(b.readBigUInt64LE(o));
    parts.push(b.slice(o + 8, o + 8 + n));
    o += 8 + n;
  }
  if (o !== b.length) throw new Error('ProverOutput: trailing bytes');
  return { proof: parts[0], programOutputs: parts[1], publicInputs: parts[2] };
}
// ConstraintComputeResult { frag_index, frag_num, constraint_evaluations: Vec<Vec<Felt>> }  (utils.rs:417-422)
function decodeConstraintResult(buf) {
  const b = Buffer.from(buf.buffer, buf.byteOffset, buf.length);
  const fragIndex = Number(b.readBigUInt64LE(0)), fragNum = Number(b.readBigUInt64LE(8)), ncols = Number(b.readBigUInt64LE(16));
  const columns = [];
  let o = 24;
  for (let c = 0; c < ncols; c++) {
    const n = Number(b.readBigUInt64LE(o));
    o += 8;
    const col = new BigUint64Array(n);
    for (let i = 0; i < n; i++) col[i] = b.readBigUInt64LE(o + 8 * i);
    o += 8 * n;
    columns.push(col);
  }
  if (o !== b.length) throw new Error('ConstraintComputeResult: trailing bytes');
  return { fragIndex, fragNum, columns };
}

module.exports = { encodeHashingWorkItem, decodeHashingResult, decodeProverOutput, decodeConstraintResult };
