// Golden-vector generator (test infrastructure; runs ONLY in the build container).
//
// Drives the reference's own prebuilt WebAssembly (/root/reference/wasm-build/*.wasm,
// i.e. the reference protocol layer + herumi/mcl compiled by the reference's Makefile:57-65)
// under node and records inputs + observed outputs as JSON fixtures in tests/golden/.
// Nothing from /root/reference is copied: fixtures hold only data (base64 messages,
// strings, booleans).  Usage:  node oracle/gen_fixtures.js [--curve bn254|bls12_381] [outdir]
//
// --curve bls12_381: the shipped wasm is mcl built with MCL_MAX_BIT_SIZE=384 (/root/reference/Makefile:65) and initPairing()
// (wasm-src/el-passo-rp.cc:8-10) reads mcl's DEFAULT CurveParam {const char* z; int b; int xi_a; bool isMTwist; int curveType}
// from the module's linear memory.  oracle/wasm_curve.js overwrites that struct with mcl's BLS12-381 parameters
// {"-0xd201000000010000", 4, 1, true, MCL_BLS12_381 = 5} before initPairing() runs; every byte of code that then executes is
// the reference's own (protocol layer + mcl).  The patch is asserted: a G1 must serialise to 48 bytes afterwards.
//
// API reachable from JS: wasm-src/el-passo-idp.cc:13-55, el-passo-user.cc:26-98,
// el-passo-rp.cc:12-40, tests.cc:100-105 (run_tests).
'use strict';
const fs = require('fs');
const path = require('path');
const { REF, selectCurve, curveFromArgv } = require('./wasm_curve.js');
const { curve: CURVE, rest: ARGV } = curveFromArgv(process.argv.slice(2));
const OUT = ARGV[0] || path.join(__dirname, '..', 'tests', 'golden');
const FB = CURVE === 'bls12_381' ? 48 : 32;              // serialised Fp size
const PREFIX = CURVE === 'bls12_381' ? 'bls12_381' : 'bn254';
const CURVE_LABEL = CURVE === 'bls12_381' ? 'BLS12_381(mcl CurveParam patched into the reference wasm)' : 'BN254(mcl default)';

function load(n) {
  return new Promise(r => {
    const M = require(REF + n);
    M.onRuntimeInitialized = () => { selectCurve(M, CURVE); M.initPairing(); r(M); };
  });
}

function b64(buf) { return Buffer.from(buf).toString('base64'); }
function unb64(s) { return Buffer.from(s, 'base64'); }

// IdProof TLV (src/ps-encoding.cc:451-489): 01 F sig1 | 01 F sig2 | 02 2F k | 01 F phi | 03 20 c | ...   (F = 32 / 48)
const OFF = { sig1: 2, sig2: FB + 4, k: 2 * FB + 6, phi: 4 * FB + 8, c: 5 * FB + 10 };

function mutate(proofB64, kind) {
  const b = unb64(proofB64);
  switch (kind) {
    case 'flip_c_bit0': b[OFF.c] ^= 1; break;
    case 'flip_sig1_ysign': b[OFF.sig1 + FB - 1] ^= 0x80; break;
    case 'flip_k_ysign': b[OFF.k + 2 * FB - 1] ^= 0x80; break;
    case 'flip_phi_ysign': b[OFF.phi + FB - 1] ^= 0x80; break;
    case 'sig1_zero': b.fill(0, OFF.sig1, OFF.sig1 + FB); break;
    case 'sig_both_zero': b.fill(0, OFF.sig1, OFF.sig1 + FB); b.fill(0, OFF.sig2, OFF.sig2 + FB); break;
    case 'flip_r0_bit0': b[OFF.c + 32 + 2 + 1] ^= 1; break;   // 06 m 20 r0...
    default: throw new Error(kind);
  }
  return b64(b);
}

(async () => {
  const idp = await load('el-passo-idp.js');
  const usr = await load('el-passo-user.js');
  const rp = await load('el-passo-rp.js');

  const scenarios = [
    { name: 'A3H2', A: 3, H: 2, nproofs: 4 },
    { name: 'A8H4', A: 8, H: 4, nproofs: 6 },
    { name: 'A16H4', A: 16, H: 4, nproofs: 2 },
    { name: 'A16H8', A: 16, H: 8, nproofs: 2 },
    { name: 'A2H1', A: 2, H: 1, nproofs: 2 },
    { name: 'A1H1', A: 1, H: 1, nproofs: 2 },
  ];
  const svcNames = ['svc', 'service', 'rp.example.org', 'a', 'b', 'c', 'd', 'e', 'f', 'g', 'h', 'i', 'j',
    'k', 'l', 'm', 'n', 'o', 'p', 'q', 'r', 's', 't', 'u', 'v', 'w', 'x', 'y', 'z', 'abc', 'ghi', 'jkl'];

  const out = { curve: CURVE_LABEL, generator: 'oracle/gen_fixtures.js --curve ' + CURVE, scenarios: [] };
  for (const sc of scenarios) {
    const S = new idp.PSSigner(sc.A);
    S.key_gen();
    const pk = S.get_pub_key().toBufferString().toBase64();
    // pk TLV = 01 F g | 02 2F gg | 02 2F XX | 04 n (F+1)*n Yi | 05 n (2F+1)*n YYi : proves which curve the module runs
    if (unb64(pk).length !== (FB + 2) + 2 * (2 * FB + 2) + 2 + sc.A * (FB + 1) + 2 + sc.A * (2 * FB + 1))
      throw new Error('public key size ' + unb64(pk).length + ' is not the ' + CURVE + ' size');
    const U = new usr.PSRequester(usr.PSPubKey.fromBufferString(usr.PSBuffer.fromBase64(pk)));
    const V = new rp.PSVerifier(rp.PSPubKey.fromBufferString(rp.PSBuffer.fromBase64(pk)));
    const vals = [], parts = [];
    for (let i = 0; i < sc.A; i++) {
      const v = 'attr' + i + '-' + sc.name;
      vals.push(v);
      parts.push(v, i < sc.H ? 'Y' : 'N');
    }
    const attrs = parts.join(' ');
    const ad = 'ad-' + sc.name;
    const rec = { name: sc.name, A: sc.A, H: sc.H, pk, attrs, attr_values: vals, ad, requests: [], proofs: [] };

    // --- request_id -> provide_id (IdP accept/reject) -> unblind
    let ub = null;
    for (let t = 0; t < 2; t++) {
      const req = usr.el_passo_request_id(U, attrs, ad);
      const cred = idp.el_passo_prove_id(S, req, ad);   // helper = provide_id (el-passo-idp.cc:13-25)
      const credWrongAd = idp.el_passo_prove_id(S, req, ad + 'x');
      const rb = unb64(req); rb[FB + 4] ^= 1;            // 01 F A(F) 03 20 c...: flip bit 0 of c
      const credFlipC = idp.el_passo_prove_id(S, b64(rb), ad);
      const ubc = U.unblind_credential(usr.PSCredential.fromBufferString(usr.PSBuffer.fromBase64(cred)));
      ub = ubc;
      rec.requests.push({
        request: req, accept: cred !== '', credential: cred,
        unblinded: ubc.toBufferString().toBase64(),
        wrong_ad_accept: credWrongAd !== '', request_flip_c: b64(rb), flip_c_accept: credFlipC !== '',
      });
    }
    // --- prove (no id retrieval) -> verify
    const proofsRaw = [];
    for (let t = 0; t < sc.nproofs; t++) {
      const svc = svcNames[t % svcNames.length];
      const proof = usr.el_passo_prove_id(U, ub, attrs, ad, svc);
      proofsRaw.push(proof);
      const P = rp.IdProof.fromBufferString(rp.PSBuffer.fromBase64(proof));
      const cases = [];
      const chk = (label, pB64, ad_, svc_) => {
        const PP = rp.IdProof.fromBufferString(rp.PSBuffer.fromBase64(pB64));
        cases.push({ label, proof: pB64, ad: ad_, svc: svc_, expect: V.el_passo_verify_id_without_id_retrieval(PP, ad_, svc_) });
      };
      chk('original', proof, ad, svc);
      chk('wrong_ad', proof, ad + '!', svc);
      chk('wrong_svc', proof, ad, svc + '!');
      chk('empty_ad', proof, '', svc);
      for (const k of ['flip_c_bit0', 'flip_sig1_ysign', 'flip_k_ysign', 'flip_phi_ysign', 'sig1_zero', 'sig_both_zero', 'flip_r0_bit0'])
        chk(k, mutate(proof, k), ad, svc);
      if (t > 0) {
        // signature pair taken from a different randomisation of the same credential
        const a = unb64(proof), b = unb64(proofsRaw[t - 1]);
        b.copy(a, OFF.sig1, OFF.sig1, OFF.sig2 + FB);
        chk('foreign_sig_pair', b64(a), ad, svc);
      }
      rec.proofs.push({ svc, username: rp.PSVerifier.get_user_name_from_signon_request(P), cases });
    }
    out.scenarios.push(rec);
  }

  // --- hashAndMapToG1 coverage: many service names on one key (A=2,H=1 keeps it small)
  {
    const S = new idp.PSSigner(2); S.key_gen();
    const pk = S.get_pub_key().toBufferString().toBase64();
    const U = new usr.PSRequester(usr.PSPubKey.fromBufferString(usr.PSBuffer.fromBase64(pk)));
    const V = new rp.PSVerifier(rp.PSPubKey.fromBufferString(rp.PSBuffer.fromBase64(pk)));
    const attrs = 'sec Y pub N', ad = 'h2c';
    const req = usr.el_passo_request_id(U, attrs, ad);
    const cred = idp.el_passo_prove_id(S, req, ad);
    const ub = U.unblind_credential(usr.PSCredential.fromBufferString(usr.PSBuffer.fromBase64(cred)));
    const list = [];
    for (const svc of svcNames) {
      const proof = usr.el_passo_prove_id(U, ub, attrs, ad, svc);
      const P = rp.IdProof.fromBufferString(rp.PSBuffer.fromBase64(proof));
      list.push({ svc, proof, expect: V.el_passo_verify_id_without_id_retrieval(P, ad, svc) });
    }
    out.hash_to_g1 = { pk, attrs, attr_values: ['sec', 'pub'], ad, cases: list };
  }
  fs.writeFileSync(path.join(OUT, PREFIX + '_oracle_flows.json'), JSON.stringify(out, null, 1));

  // --- with-id-retrieval flow: tests.wasm run_tests prints base64 of every message (wasm-src/tests.cc:11-94)
  const runs = [];
  const cp = require('child_process');
  // run_tests() calls initPairing() itself (wasm-src/tests.cc:100-105): the curve is selected before it
  const script = `const C=require('${path.join(__dirname, 'wasm_curve.js')}');const M=require('${REF}tests.js');` +
    `M.onRuntimeInitialized=()=>{C.selectCurve(M,'${CURVE}');M.ccall('run_tests',null,[],[]);};`;
  for (let t = 0; t < 3; t++) {
    const txt = cp.execFileSync(process.execPath, ['-e', script], { encoding: 'utf8' });
    const grab = (tag) => { const m = txt.match(new RegExp(tag + ' Base64: (\\S+)')); return m ? m[1] : null; };
    runs.push({
      pk: grab('PK'), request: grab('RequestID'), credential: grab('Credential'), proof: grab('ProveID'),
      verify_failed_line: /EL PASSO Verify ID failed/.test(txt),
      attrs: 's Y gamma Y tp N', attr_values: ['s', 'gamma', 'tp'], ad: 'hello', svc: 'service',
      g_seed: 'abc', authority_pk_seed: 'ghi', h_seed: 'jkl',
    });
  }
  fs.writeFileSync(path.join(OUT, PREFIX + '_oracle_with_retrieval.json'),
    JSON.stringify({ curve: CURVE_LABEL, generator: 'oracle/gen_fixtures.js --curve ' + CURVE, runs }, null, 1));
  console.log('fixtures written to', OUT);
})();
