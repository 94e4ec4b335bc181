// The SDK's constraint worker (aero-sdk/src/constraints_worker.ts -> miden-wasm `constraint_entry_point`, constraints_worker.rs:81-97)
// as a Node worker thread: one bincode ConstraintComputeWorkItem in, one bincode ConstraintComputeResult out. workerData.air =
// [auxWidth, auxRands, auxDegree] of the built-in AIR's auxiliary segment, or null.
'use strict';
const { parentPort, workerData } = require('worker_threads');
const aero = require('./aero_worker.node');

const handle = aero.open(workerData.lib, workerData.device || 0);
parentPort.on('message', (payload) => {
  if (payload === 'close') { aero.close(handle); parentPort.close(); return; }
  parentPort.postMessage(aero.evalConstraints(handle, payload, workerData.air || null));
});
