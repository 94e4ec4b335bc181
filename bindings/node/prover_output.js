// bincode ProverOutput of a proof container (what the proving worker posts back to the SDK, proving_worker.rs:205-222), host only.
// usage: node prover_output.js <libaero_stark.so> <container.bin>   -> JSON { hex, proofLen, programOutputsLen, publicInputsLen }
'use strict';
const fs = require('fs');
const path = require('path');
const aero = require('./aero_worker.node');
const { decodeProverOutput } = require('./bincode');

const blob = fs.readFileSync(process.argv[3]);
const n = Number(blob.readBigUInt64LE(0));
const inputs = blob.slice(8, 8 + n);
const m = Number(blob.readBigUInt64LE(8 + n));
const proof = blob.slice(16 + n, 16 + n + m);
const h = aero.open(path.resolve(process.argv[2]), -1);            // no GPU context: the encoders are host code
const out = aero.proverOutput(h, proof, inputs);
const parts = decodeProverOutput(out);
console.log(JSON.stringify({ hex: out.toString('hex'), proofLen: parts.proof.length, programOutputsLen: parts.programOutputs.length,
                             publicInputsLen: parts.publicInputs.length }));
