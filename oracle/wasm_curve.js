// Curve selection for the reference's prebuilt wasm modules (test infrastructure; build container only).
//
// The modules are the reference protocol layer + herumi/mcl compiled with -DMCL_MAX_BIT_SIZE=384
// (/root/reference/Makefile:57-65), so mcl's 381-bit code paths are inside them; what makes them run BN254 is only that
// initPairing() is called with no argument (/root/reference/wasm-src/el-passo-rp.cc:8-10, el-passo-idp.cc:9, el-passo-user.cc:11,
// tests.cc:103) and mcl's default argument is the static `mcl::BN254` CurveParam
//     struct CurveParam { const char* z; int b; int xi_a; bool isMTwist; int curveType; };
// which lives in the module's data segment.  selectCurve(M, 'bls12_381') finds that struct in linear memory (the word that
// points at the string "-0x4080000000000001", followed by b = 2 and xi_a = 1) and overwrites it with mcl's BLS12-381
// parameters { "-0xd201000000010000", 4, 1, true, MCL_BLS12_381 = 5 }.  No reference code is modified or copied: the
// module then executes its own mcl on the other curve.  Callers must assert the effect (G1 serialises to 48 bytes).
'use strict';
const REF = '/root/reference/wasm-build/';
const Z_BN254 = '-0x4080000000000001';
const Z_BLS12_381 = '-0xd201000000010000';

function findString(H8, s) {
  const b = Buffer.from(s + '\0');
  const hits = [];
  for (let i = 0; i + b.length <= H8.length; i++) {
    let ok = true;
    for (let j = 0; j < b.length; j++) if (H8[i + j] !== b[j]) { ok = false; break; }
    if (ok) hits.push(i);
  }
  return hits;
}

function selectCurve(M, curve) {
  if (curve === 'bn254') return null;
  if (curve !== 'bls12_381') throw new Error('unknown curve ' + curve);
  const H8 = M.HEAPU8, H32 = M.HEAPU32;
  const structs = [];
  for (const z of findString(H8, Z_BN254))
    for (let i = 0; i + 5 <= H32.length; i++)
      if (H32[i] === z && H32[i + 1] === 2 && H32[i + 2] === 1) structs.push(i);
  if (structs.length !== 1) throw new Error('default CurveParam not found exactly once: ' + structs.length);
  const w = structs[0];
  const zs = Buffer.from(Z_BLS12_381 + '\0');
  const p = M._malloc(zs.length + 8);
  M.HEAPU8.set(zs, p);                       // (re-read the views: _malloc may have grown the heap)
  const V32 = M.HEAPU32, V8 = M.HEAPU8;
  V32[w] = p; V32[w + 1] = 4; V32[w + 2] = 1; V8[4 * (w + 3)] = 1; V32[w + 4] = 5;
  return 4 * w;
}

function curveFromArgv(argv) {
  let curve = 'bn254';
  const rest = [];
  for (let i = 0; i < argv.length; i++) {
    if (argv[i] === '--curve') curve = String(argv[++i]).toLowerCase();
    else rest.push(argv[i]);
  }
  if (curve !== 'bn254' && curve !== 'bls12_381') throw new Error('--curve bn254|bls12_381');
  return { curve, rest };
}

module.exports = { REF, selectCurve, curveFromArgv };
