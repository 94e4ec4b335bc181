// The SDK's hashing worker (aero-sdk/src/hashing_worker.ts -> miden-wasm `hashing_entry_point`, hashing_worker.rs:28-42) as a Node
// worker thread backed by the GPU library: one bincode HashingWorkItem in, one bincode HashingResult out, one context per worker.
'use strict';
const { parentPort, workerData } = require('worker_threads');
const aero = require('./aero_worker.node');

const handle = aero.open(workerData.lib, workerData.device || 0);
parentPort.on('message', (payload) => {
  if (payload === 'close') { aero.close(handle); parentPort.close(); return; }
  parentPort.postMessage(aero.hashRows(handle, payload));
});
