// Asks the reference's prebuilt modules (wasm-build/el-passo-{rp,idp}.{js,wasm} = reference protocol layer + mcl) for their verdict on a
// list of cases read as JSON from stdin; prints the verdicts (true / false / "throw:<msg>") as JSON.
//   {pk, proof, ad, svc}          -> PSVerifier::el_passo_verify_id_without_id_retrieval (RP module, wasm-src/el-passo-rp.cc:12-40)
//   {kind: "request", sk.., ...}  -> not reachable: the IdP's secret key never leaves its module, so request verdicts are taken by
//                                    oracle/gen_fixtures.js inside the module that generated the key
// Usage: node oracle/wasm_verify.js [--curve bn254|bls12_381] < cases.json     (curve selection: oracle/wasm_curve.js)
// Test infrastructure, build container only (used by oracle/gen_edge_fixtures.py and by anyone re-verifying tests/golden/*.json).
'use strict';
const { REF, selectCurve, curveFromArgv } = require('./wasm_curve.js');
const { curve } = curveFromArgv(process.argv.slice(2));
const rp = require(REF + 'el-passo-rp.js');
let input = '';
process.stdin.on('data', d => input += d);
process.stdin.on('end', () => {
  const go = () => {
    selectCurve(rp, curve);
    rp.initPairing();
    const cases = JSON.parse(input);
    const out = [];
    const FB = curve === 'bls12_381' ? 48 : 32;
    for (const c of cases) {
      try {
        const pkb = Buffer.from(c.pk, 'base64');
        if (pkb[1] !== FB) throw new Error('public key is not a ' + curve + ' key');
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
