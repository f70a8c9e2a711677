// Asks the reference's prebuilt RP module (wasm-build/el-passo-rp.{js,wasm} = reference protocol layer + mcl) for its verdict on a
// list of (pk, proof, ad, svc) cases read as JSON from stdin; prints the verdicts (true / false / "throw:<msg>") as JSON.
// Test infrastructure, build container only (used by oracle/gen_edge_fixtures.py).
'use strict';
const REF = '/root/reference/wasm-build/';
const rp = require(REF + 'el-passo-rp.js');
let input = '';
process.stdin.on('data', d => input += d);
process.stdin.on('end', () => {
  const go = () => {
    rp.initPairing();
    const cases = JSON.parse(input);
    const out = [];
    for (const c of cases) {
      try {
        const V = new rp.PSVerifier(rp.PSPubKey.fromBufferString(rp.PSBuffer.fromBase64(c.pk)));
        const P = rp.IdProof.fromBufferString(rp.PSBuffer.fromBase64(c.proof));
        out.push(V.el_passo_verify_id_without_id_retrieval(P, c.ad, c.svc));
      } catch (e) {
        out.push('throw:' + String(e).slice(0, 80));
      }
    }
    console.log(JSON.stringify(out));
  };
  if (rp.calledRun) go(); else rp.onRuntimeInitialized = go;
});
