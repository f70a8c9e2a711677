// IdP-side golden vectors for requests the reference's own requester never emits (test infrastructure; build container only).
//
// The IdP's secret key cannot leave the reference's wasm module (wasm-src/el-passo-idp.cc:13-55 exposes key_gen, get_pub_key and
// el_passo_provide_id only), so this driver keeps the module's PSSigner alive, has oracle/gen_request_edge.py (big-int model) build
// requests for that key -- the model's own el_passo_request_id, and commitments outside G1 on BLS12-381 -- and records what
// PSSigner::el_passo_provide_id (src/ps-signer.cc:63-72) answers: accept / reject and the blind signature.
// Output: tests/golden/<curve>_oracle_requests.json (data only).  Usage: node oracle/gen_request_fixtures.js [--curve bn254|bls12_381] [outdir]
'use strict';
const fs = require('fs');
const path = require('path');
const cp = require('child_process');
const { REF, selectCurve, curveFromArgv } = require('./wasm_curve.js');
const { curve: CURVE, rest: ARGV } = curveFromArgv(process.argv.slice(2));
const OUT = ARGV[0] || path.join(__dirname, '..', 'tests', 'golden');

const idp = require(REF + 'el-passo-idp.js');
idp.onRuntimeInitialized = () => {
  selectCurve(idp, CURVE);
  idp.initPairing();
  const out = { curve: CURVE === 'bls12_381' ? 'BLS12_381(mcl CurveParam patched into the reference wasm)' : 'BN254(mcl default)',
    generator: 'oracle/gen_request_fixtures.js --curve ' + CURVE + ' + oracle/gen_request_edge.py', scenarios: [] };
  for (const sc of [{ name: 'A3H2', A: 3, H: 2 }, { name: 'A8H4', A: 8, H: 4 }, { name: 'A2H1', A: 2, H: 1 }]) {
    const S = new idp.PSSigner(sc.A);
    S.key_gen();
    const pk = S.get_pub_key().toBufferString().toBase64();
    const attrs = [];
    for (let i = 0; i < sc.A; i++) attrs.push(['attr' + i + '-' + sc.name, i < sc.H]);
    const ad = 'req-' + sc.name;
    const made = JSON.parse(cp.execFileSync('python3', [path.join(__dirname, 'gen_request_edge.py'), '--curve', CURVE],
      { input: JSON.stringify({ pk, attrs, ad, seed: 20213 + sc.A }), encoding: 'utf8' }));
    const cases = [];
    for (const m of made) {
      const cred = idp.el_passo_prove_id(S, m.request, ad);       // helper = provide_id (el-passo-idp.cc:13-25)
      const credWrongAd = idp.el_passo_prove_id(S, m.request, ad + 'x');
      cases.push({ label: m.label, request: m.request, t1: m.t1, accept: cred !== '', credential: cred, wrong_ad_accept: credWrongAd !== '' });
      console.log(sc.name, m.label, cred !== '');
    }
    out.scenarios.push({ name: sc.name, A: sc.A, H: sc.H, pk, attr_values: attrs.map(a => a[0]), ad, cases });
  }
  fs.writeFileSync(path.join(OUT, (CURVE === 'bls12_381' ? 'bls12_381' : 'bn254') + '_oracle_requests.json'), JSON.stringify(out, null, 1));
};
