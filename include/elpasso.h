/*
 * elpasso.h — C-ABI of the MI355X-native (HIP, gfx950) hot path under the PS-signature / EL PASSO protocol layer.
 *
 * The reference (Zhiyi-Zhang/PS-Signature-and-EL-PASSO) has no FFI boundary: src/ps-*.cc call herumi/mcl's C++ API
 * directly.  This header is the boundary a maintainer would bind instead of mcl for the batched path;
 * each entry point cites the reference code it replaces.  Plain C types only; the caller owns every buffer; no
 * exceptions cross the ABI (int status, per-item results in uint8_t flag arrays).  One elp_ctx per host thread.
 *
 * Encodings ("std" = canonical integers, little-endian; F = 32 bytes for BN254, 48 for BLS12-381):
 *   Fr        : 32 bytes                                      (mcl Fr::serialize)
 *   G1 affine : x[F] | y[F]              all-zero = infinity  (uncompressed; mcl wire form is x with a parity flag)
 *   G2 affine : x.a[F] | x.b[F] | y.a[F] | y.b[F]
 *   G1 wire   : F bytes, G2 wire: 2F bytes                    (mcl G1/G2::serialize, src/ps-encoding.cc:167,199)
 *   GT        : 12 x F bytes, tower order c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (each a | b)
 * All buffers must be 4-byte aligned.
 */
#ifndef ELPASSO_H_
#define ELPASSO_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct elp_ctx elp_ctx;

enum { ELP_CURVE_BN254 = 0, ELP_CURVE_BLS12_381 = 1 };
enum {
  ELP_OK = 0,
  ELP_ERR_ARG = -1,      /* bad argument */
  ELP_ERR_HIP = -2,      /* HIP runtime error (see elp_last_error) */
  ELP_ERR_STATE = -3,    /* required key material not set */
  ELP_ERR_NODEVICE = -4, /* no usable GPU: there is NO CPU fallback */
  ELP_ERR_POINT = -5     /* a key/base point is not on the curve */
};

/* ---- context ------------------------------------------------------------------------------------------------ */
/* replaces mcl::bls12::initPairing() (test/ps-tests.cc:142): selects the curve and the GPU. */
int elp_init(int curve, int device, elp_ctx** out);
/* GPUs visible to the process (hipGetDeviceCount; 0 without one): the ordinals elp_init accepts.  The host layer's PSVerifier(pk, devices, W) shards batches
 * over one context per listed device. */
int elp_device_count(void);
void elp_destroy(elp_ctx* ctx);
const char* elp_last_error(const elp_ctx* ctx);
int elp_field_bytes(int curve);               /* F */
/* Options (elp_set_option).  Every option except the two POLICY options selects among kernels whose verdicts are identical; which kernel serves which
 * batch size, and the measurements behind each default, are in DESIGN.md section 5 and profiles/option_history.md.
 *
 * POLICY (what is accepted):
 * ELP_OPT_STRICT_SIGNATURE (default 1): el_passo_verify_id rejects a proof whose sig1 is the point at infinity and, on BLS12-381 with
 *   ELP_OPT_SUBGROUP_CHECK on, a sig1 outside G1.  The reference has no such test in el_passo_verify_id (src/ps-verifier.cc:133-137): it accepts
 *   sig1 = sig2 = infinity with a self-made NIZK (golden case "sig_both_zero", both curves) and, run on BLS12-381, (sig1, sig2) = (T, O) with T of
 *   order 3 as well as sig1 + T (tests/golden/bls12_381_oracle_edge.json "sig_T3_O", "sig1_plus_T3") -- universal forgeries, since a point whose
 *   order divides the cofactor pairs to 1 with everything.  0 = the reference's behaviour bit for bit (together with ELP_OPT_SUBGROUP_CHECK = 0 on
 *   BLS12-381).  elp_ps_verify_batch always rejects sig1 = infinity as PSVerifier::verify does (src/ps-verifier.cc:16-18) and, with the subgroup check on,
 *   sig1 outside G1.
 * ELP_OPT_SUBGROUP_CHECK (default 1; BLS12-381 only -- E(Fp) of BN254 has prime order): the prover-supplied G1 points of a proof or request -- phi, E1,
 *   E2, the commitment A of el_passo_provide_id, and sig1 where the strict rule applies -- must lie in the order-r subgroup, otherwise the item is
 *   rejected.  A DELIBERATE divergence from the reference: mcl's default does not test the order of a deserialised G1 point.  What the reference's own
 *   wasm answers when run on this curve (tests/golden/bls12_381_oracle_edge.json, _requests.json): phi + T with an honest transcript: rejected; a crafted
 *   phi + T3 or A + T3 with 3 | c: accepted iff mcl's GLV split c = a + b (z^2 - 1) has 3 | a + b (about one in three); sig1 / sig2 + T: accepted.  phi is
 *   the user's pseudonym and (E1, E2) the identity-retrieval token: a small-order component would give one user several pseudonyms or an undecryptable
 *   token, hence the check.  0 = no check, on EVERY path: for inputs the caller has validated.  The sig1 cases are then accepted exactly as the reference
 *   accepts them; the verdict on a phi / E1 / E2 / A outside G1 is implementation-defined (the device's GLV split is not mcl's), and (sig1 of order 3,
 *   sig2 = O) verifies.
 *
 * KERNEL SELECTION (verdicts identical):
 * ELP_OPT_PAIRED_LAYOUT: 0 = one lane per item; 1 = two lanes per item (the Fp2 tower split over a lane pair; BN254 builds); 2 (default) = by batch
 *   size: two lanes when the last round of 64 x SIMDs items would be at most half full and always on BLS12-381, one lane otherwise.  On BLS12-381 the
 *   main kernel of elp_verify_id_batch_aggregated[_dev] follows the option too (two lanes unless the value is 0; on BN254 it is a one-lane kernel).
 * ELP_OPT_TABLE_WORKSPACE (default 1): the per-item tables of the variable-base multiplications live in a launch workspace in device memory (3 KB per
 *   item, see the *_dev entry points) instead of the lanes' private memory; 0 saves the memory at a few per cent of throughput.
 * ELP_OPT_SPLIT_PHASES (default 0): 1 = the one-lane el_passo_verify_id as two kernels (NIZK half as independent jobs, then the pairing check; BN254);
 *   2 = the G2 job and the G1 jobs as concurrent kernels on two streams, then the pairing check (BN254); 3 = the base-field jobs (V_phi, V_E1, V_E2 and,
 *   on BLS12-381, the subgroup tests) as a kernel of their own in front of the two-lane kernel (both curves).
 * ELP_OPT_COOP_PAIRING (default 1): small batches -- PS verifications of at most 4096 items, el_passo_verify_id of at most 9216 (BLS12-381: 8192);
 *   value > 1: that many for both -- and the closing step of aggregated verification run the pairing check cooperatively (32 lanes per item, an Fp2
 *   register file in LDS, a level-scheduled program, csrc/elp/coop.h), with the NIZK half spread over job waves in the same launch.  0 = off.
 * ELP_OPT_COALESCED_RECORDS (default 1; BN254, plain layout): the 64 records of a workgroup are fetched as one contiguous block through LDS
 *   (k_verify_id_staged) instead of being read in place.  Needs records of a multiple of 16 bytes, at most 1152, 16-byte aligned; otherwise the
 *   in-place kernel runs.
 * ELP_OPT_STREAM_OVERLAP (default 0): independent kernels of ONE call run on a second stream owned by the context, joined by events before the call's
 *   last kernel; the caller's stream semantics are unchanged.  Only kernels with small private frames go there (scratch memory is provisioned per
 *   hardware queue; profiles/r04_scratch_stall.md).  Rule for callers: launch the verification batches of one process from ONE stream (two at most).
 * ELP_OPT_PAIR4 (default 1): the pairing check on FOUR lanes per item (a DPP quad; csrc/elp/quad.h, pair4.h) for el_passo_verify_id of 3 073 ... 16 384
 *   items (k_vid_mid) and PS verifications of 4 097 ... 16 384 items (k_pair4).  0 = off, 2 = wherever the path exists (up to 131 072 items; A/B runs).
 * ELP_OPT_WIRE_DECODE (default 1): elp_verify_id_wire_batch[_dev] with at most 16 384 messages decodes them into records first (k_wire_decode) and
 *   verifies the records on the small / mid-size paths when all messages hide the same attributes; mixed patterns and larger batches take the fused wire
 *   kernels.  NOTE: the decision needs one 16-byte read-back, so on this path the _dev entry point SYNCHRONISES the caller's stream once inside the call
 *   (it does not overlap with work queued behind it), and a single message with another hidden pattern sends the whole batch to the fused kernels
 *   (about 3x slower at these sizes): callers who cannot trust their senders to use one pattern should group messages by pattern or set 0.
 * ELP_OPT_AGG_TWO_PER_LANE (default 0; BN254): elp_verify_id_batch_aggregated[_dev] with two proofs on a lane (shared squarings).  1 = on batches that
 *   need fewer rounds of lanes that way, 2 = always (tests).  Off by default: largest private frame of the library (scratch re-provisioning).
 * ELP_OPT_PAIR16 (default 1; round 6): PS verifications of at most 4 096 items (value > 1: that many) run the pairing check with ONE ITEM PER 16-LANE ROW of a wave --
 *   12 lanes hold one base-field coefficient each of the Fp12 value, every Fp12-level operation is one inner product per lane over operands published in LDS
 *   (csrc/elpasso_pair16.h; tables and program generated and simulated by tools/gen_row16.py for both curves) -- from the size at which that wins: 4 items on BN254,
 *   2 049 on BLS12-381 (below, the cooperative interpreter's 32 lane pairs per item are faster); on both curves also the closing step of aggregated verification
 *   (k_agg_final16) and the product of its per-wave Miller values (k_fp12_reduce16: every product one step on a row).
 *   0 = the interpreter and the one-lane product tree keep these jobs.  Measurements: profiles/r06_pair16.md.
 * ELP_OPT_FAULT_INJECT (default 0; test hook for callers' error paths): the next `value` calls of elp_verify_id_batch_submit on this context fail with
 *   ELP_ERR_STATE before anything is queued.  No other entry point consumes or honours the counter.
 */
enum { ELP_OPT_STRICT_SIGNATURE = 1, ELP_OPT_PAIRED_LAYOUT = 2, ELP_OPT_TABLE_WORKSPACE = 3, ELP_OPT_SPLIT_PHASES = 4, ELP_OPT_SUBGROUP_CHECK = 5,
       ELP_OPT_COOP_PAIRING = 6, ELP_OPT_COALESCED_RECORDS = 7, ELP_OPT_STREAM_OVERLAP = 8, ELP_OPT_FAULT_INJECT = 9, ELP_OPT_PAIR4 = 10, ELP_OPT_WIRE_DECODE = 11,
       ELP_OPT_AGG_TWO_PER_LANE = 12, ELP_OPT_PAIR16 = 13 };
int elp_set_option(elp_ctx* ctx, int option, int value);
const char* elp_version(void);

/* ---- key material (builds fixed-base window tables and the Miller-loop lines of gg in HBM) ------------------ */
/* PSPubKey{g, gg, XX, Yi[A], YYi[A]} (src/ps-encoding.h:111-140) as affine std points. window_bits 0 = default (8), up to 22
 * (signed digits: table bytes grow as 2^(W-1) / W: 1.25 GiB at W = 16, 16 GiB at W = 20 for an 8-attribute BN254 key; elp_key_table_bytes). */
int elp_set_pubkey(elp_ctx* ctx, int nattr, const uint8_t* g, const uint8_t* gg, const uint8_t* XX, const uint8_t* Yi,
                   const uint8_t* YYi, int window_bits);
/* RP parameters of el_passo_verify_id (src/ps-verifier.h:45-49): service name (hashAndMapToG1 is evaluated on the GPU),
 * ElGamal authority_pk, g, h (G1 affine std; NULL when id-retrieval is not used). */
int elp_set_rp(elp_ctx* ctx, const uint8_t* service_name, size_t service_len, const uint8_t* authority_pk,
               const uint8_t* g, const uint8_t* h);
/* PSSigner's secret X = g^x (src/ps-signer.h:94), G1 affine std. */
int elp_set_signer_secret(elp_ctx* ctx, const uint8_t* X);

/* ---- batched primitives over host buffers (n independent items) ---------------------------------------------- */
/* G1/G2::deserialize (point decompression; src/ps-encoding.cc:192,224): wire -> affine std; ok[i] = 0 if invalid */
int elp_g1_decompress(elp_ctx* ctx, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
int elp_g2_decompress(elp_ctx* ctx, size_t n, const uint8_t* wire, uint8_t* out, uint8_t* ok);
/* G1::mul / G2::mul, variable base (e.g. src/ps-verifier.cc:73,92; src/ps-requester.cc:109,144-146) */
int elp_g1_mul(elp_ctx* ctx, size_t n, const uint8_t* points, const uint8_t* scalars, uint8_t* out);
int elp_g2_mul(elp_ctx* ctx, size_t n, const uint8_t* points, const uint8_t* scalars, uint8_t* out);
/* G1::add / G2::add (src/ps-verifier.cc:28,81) */
int elp_g1_add(elp_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
int elp_g2_add(elp_ctx* ctx, size_t n, const uint8_t* a, const uint8_t* b, uint8_t* out);
/* sum_t scalars[i][t] * base[base_ids[t]] over the key's fixed bases (loops of G?::mul + G?::add over pk.Yi / YYi / g /
 * gg / XX, src/ps-verifier.cc:76-88,220-227; src/ps-signer.cc:88-94,121-128).
 * G1 base ids: 0 = g, 1+i = Y_i, A+1 = H1(service), A+2 = g_elgamal, A+3 = authority_pk, A+4 = h, A+5 = X.
 * G2 base ids: 0 = gg, 1 = XX, 2+i = YY_i. */
int elp_g1_msm_fixed(elp_ctx* ctx, size_t n, int nterms, const int32_t* base_ids, const uint8_t* scalars, uint8_t* out);
int elp_g2_msm_fixed(elp_ctx* ctx, size_t n, int nterms, const int32_t* base_ids, const uint8_t* scalars, uint8_t* out);
/* Single-output multi-scalar multiplication  out = sum_i scalars[i] * points[i]  over caller-supplied points (Pippenger bucket
 * method: LDS counting sort by window digit, bucket sums, shuffle/LDS scan reductions).  No call in the reference corresponds to
 * it (mcl's mulVec is unused there); it is the general MSM operator and the building block of aggregated verification. */
int elp_g1_msm(elp_ctx* ctx, size_t n, const uint8_t* points, const uint8_t* scalars, uint8_t* out);
int elp_g2_msm(elp_ctx* ctx, size_t n, const uint8_t* points, const uint8_t* scalars, uint8_t* out);
/* The same multi-scalar multiplication over DEVICE buffers, asynchronous on `stream` (hipStream_t; NULL = default stream): d_points n affine std points, d_scalars
 * n x 32 bytes, d_out one affine std point, d_workspace elp_msm_workspace_bytes(curve, group, n) bytes of device memory owned by the caller (group 1 = G1, 2 = G2)
 * and untouched by others until the result is complete.  Nothing is copied or synchronised.  A point that is not on the curve contributes nothing (the host-buffer
 * entry points report it as ELP_ERR_POINT; here the caller validates). */
size_t elp_msm_workspace_bytes(int curve, int group, size_t n);
int elp_g1_msm_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
int elp_g2_msm_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_points, const void* d_scalars, void* d_workspace, void* d_out);
/* hashAndMapToG1 (src/ps-verifier.cc:94,186; src/ps-requester.cc:185,336) as mcl evaluates it: Fp::setHashOf (SHA-256 on BN254, SHA-512 on BLS12-381),
 * the Shallue-van de Woestijne map, and on BLS12-381 the cofactor (z-1)^2/3; pinned on both curves by reference-made proofs. msgs concatenated, offsets[n+1]. */
int elp_hash_to_g1(elp_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* offsets, uint8_t* out);
/* pairing(GT&, G1, G2) (src/ps-verifier.cc:32-33) */
int elp_pairing(elp_ctx* ctx, size_t n, const uint8_t* g1, const uint8_t* g2, uint8_t* gt);
/* prod_{j<npairs} e(P[i][j], Q[i][j]) == 1, one final exponentiation per item (npairs = 1..4) */
int elp_pairing_check(elp_ctx* ctx, size_t n, int npairs, const uint8_t* g1, const uint8_t* g2, uint8_t* ok);

/* ---- fused protocol batches ------------------------------------------------------------------------------------ */
/* Record sizes in bytes. H = popcount(hidden_mask) */
size_t elp_verify_id_record_size(int curve, int nattr, int nhidden, int with_retrieval);
size_t elp_ps_verify_record_size(int curve, int nattr);
size_t elp_provide_id_record_size(int curve, int nattr, int nhidden);

/* PSVerifier::el_passo_verify_id / _without_id_retrieval (src/ps-verifier.cc:37-138,140-212).
 * record i: sig1 | sig2 | phi | [E1 | E2] | k | c | rs[H+2 or H+1] | m[A-H]   (G1,G1,G1,[G1,G1],G2,Fr,...)
 *   hidden_mask bit j set <=> proof.attributes[j] == "" ; m = Fr::setHashOf(attribute) of the revealed ones, in order.
 * associated data: ad_off == NULL -> the same ad[0..ad_len) for every item, else item i uses ad[ad_off[i]..ad_off[i+1]).
 * flags[i] = 1 iff the reference would return true; *accepted = number of ones. */
int elp_verify_id_batch(elp_ctx* ctx, size_t n, const uint8_t* records, uint64_t hidden_mask, int with_retrieval,
                        const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, uint8_t* flags, uint64_t* accepted);
/* The same call in two halves for callers that PIPELINE batches (round 4): _submit queues the host-to-device copies (on the context's copy stream), the kernel and the
 * copy of the verdicts back, and returns; _wait blocks until the batch of that slot is done and delivers its accepted count.  Two slots (0, 1), each with its own device
 * buffers: while the kernel of slot 0 runs, the caller packs and submits slot 1, whose records travel over PCIe meanwhile -- a steady stream of batches then costs the kernel
 * time alone (PSVerifier::el_passo_verify_id_submit / _collect; bench.py host_api.*.objects_pipelined).  records / ad / ad_off / flags must stay untouched until _wait returns and
 * should be page-locked (elp_host_alloc), otherwise the copies are staged by the runtime and _submit blocks for their duration.  A slot must be waited for before it is
 * submitted again; verdict semantics are those of elp_verify_id_batch. */
int elp_verify_id_batch_submit(elp_ctx* ctx, int slot, size_t n, const uint8_t* records, uint64_t hidden_mask, int with_retrieval, const uint8_t* ad,
                               const uint32_t* ad_off, size_t ad_len, uint8_t* flags);
int elp_verify_id_batch_wait(elp_ctx* ctx, int slot, uint64_t* accepted);
/* Records handed over IN PARTS (round 6): a caller that packs a batch chunk by chunk (PSVerifier::el_passo_verify_id_batch: packing + attribute hashing on host threads)
 * stages every chunk as soon as it is packed -- the copy of records [first, first + count) of a batch of n_total records of record_size bytes is queued on the context's
 * copy stream and the call returns --, so that the records cross PCIe while the next chunk is being packed; elp_verify_id_batch_submit with records == NULL then
 * launches over what was staged (exactly n_total records must have been delivered; parts may arrive in any order, each byte once).  The parts must stay untouched
 * until _wait returns and should be page-locked (elp_host_alloc).  n_total x record_size must not grow between the first part and the submit. */
int elp_verify_id_batch_stage(elp_ctx* ctx, int slot, size_t n_total, size_t record_size, size_t first, size_t count, const uint8_t* records_part);
/* The same verification straight from the reference's wire messages: msgs = concatenated IdProof::toBufferString() bytes
 * (src/ps-encoding.cc:451-467; base64 already removed), message i = msgs[msg_off[i] .. msg_off[i+1]).  T-L-V parsing, point
 * decompression (G?::deserialize, src/ps-encoding.cc:192,224) and Fr::setHashOf of the revealed attributes
 * (src/ps-verifier.cc:224) all run on the GPU; the hidden pattern is taken per message from its "" placeholders.
 * Malformed / truncated messages, scalars >= r, or an attribute count different from the key's are rejected (flag 0). */
int elp_verify_id_wire_batch(elp_ctx* ctx, size_t n, const uint8_t* msgs, const uint32_t* msg_off, int with_retrieval,
                             const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, uint8_t* flags, uint64_t* accepted);
/* Aggregated variant of elp_verify_id_batch (no counterpart in the reference).  The NIZK half runs per
 * item as usual; the signature checks of the items that pass it are combined with verifier-chosen 128-bit multipliers d_i
 * (SHA-256(seed || le64(i)); seed32 == NULL (recommended): the library draws the 32-byte seed from the OS CSPRNG per call; a
 * caller-supplied seed32 must point to exactly 32 bytes that no prover can predict before the batch is fixed):
 *     prod_i e(d_i sig1_i, K_i) * e(-sum_i d_i sig2_i, gg) == 1
 * i.e. one Miller loop per item, a Pippenger MSM for sum d_i sig2_i, ONE final exponentiation per batch.  If the batch equation
 * fails the items are re-verified one by one inside the same call, so flags[] are the reference's verdicts either way (a wrong
 * accept requires a 2^-128 event).  *batch_equation_held reports which path produced them. */
int elp_verify_id_batch_aggregated(elp_ctx* ctx, size_t n, const uint8_t* records, uint64_t hidden_mask, int with_retrieval,
                                   const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, const uint8_t* seed32, uint8_t* flags,
                                   uint64_t* accepted, int* batch_equation_held);
/* PSVerifier::verify (src/ps-verifier.cc:13-35). record i: sig1 | sig2 | m[nattr] */
int elp_ps_verify_batch(elp_ctx* ctx, size_t n, const uint8_t* records, int nattr, uint8_t* flags, uint64_t* accepted);
/* PSSigner::el_passo_provide_id (src/ps-signer.cc:63-146). record i: A | c | rs[H+1] | m[A-H] | u  (u = the nonce that
 * sign_commitment draws with setByCSPRNG, src/ps-signer.cc:135-136, injected for reproducibility).
 * sigs[i] = sig1 | sig2 (zeros when rejected). */
int elp_provide_id_batch(elp_ctx* ctx, size_t n, const uint8_t* records, uint64_t hidden_mask, const uint8_t* ad,
                         const uint32_t* ad_off, size_t ad_len, uint8_t* sigs, uint8_t* flags, uint64_t* accepted);

/* ---- user side in batch ---------------------------------------------------------------
 * The reference's requester draws its randomness with Fr::setByCSPRNG; here it is an input, in the reference's draw order,
 * so a batch is reproducible (load generators, test-vector production) and comparable with the oracle bit for bit.
 * Scalars are 32-byte little-endian values < r; m[i] = Fr::setHashOf(attribute i) for ALL attributes. */
size_t elp_request_id_record_size(int curve, int nattr, int nhidden);
size_t elp_request_id_out_size(int curve, int nhidden);
size_t elp_prove_id_record_size(int curve, int nattr, int nhidden, int with_retrieval);
/* PSRequester::el_passo_request_id (src/ps-requester.cc:19-99).  record i: m[A] | t | rho_0 | rho[H]
 * requests[i] = A | c | rs[H+1]  (the head of the elp_provide_id_batch record: append m_revealed and the nonce u to issue). */
int elp_request_id_batch(elp_ctx* ctx, size_t n, const uint8_t* records, uint64_t hidden_mask, const uint8_t* ad,
                         const uint32_t* ad_off, size_t ad_len, uint8_t* requests);
/* PSRequester::el_passo_prove_id / _without_id_retrieval (src/ps-requester.cc:150-432), credential randomisation included
 * (:163-170).  record i: sig1 | sig2 | m[A] | t | r | [eps] | rho[H] | rho_t | [rho_eps]   ([..] only with retrieval).
 * proofs[i] is the elp_verify_id_batch record (elp_verify_id_record_size bytes): prover output = verifier input.
 * flags[i] = 0 (proof zeroed) when the credential points are not on the curve or attribute 0 (and 1) is not hidden. */
int elp_prove_id_batch(elp_ctx* ctx, size_t n, const uint8_t* records, uint64_t hidden_mask, int with_retrieval,
                       const uint8_t* ad, const uint32_t* ad_off, size_t ad_len, uint8_t* proofs, uint8_t* flags,
                       uint64_t* produced);

/* Same fused batches over DEVICE buffers, asynchronous on `stream` (hipStream_t; NULL = default stream).  Nothing is
 * copied or synchronised; *d_accepted (uint64 in device memory) is atomically incremented.
 * The verify_id entry points keep a device workspace per stream they were called on (the per-item tables of the variable-base
 * multiplications: 3 KB per item of the largest batch seen, e.g. 201 MB for 65 536 items), grown on demand -- growing it
 * synchronises the device once -- and freed by elp_destroy.
 * elp_verify_id_wire_batch_dev with at most 16 384 messages and ELP_OPT_WIRE_DECODE on is the one exception to "nothing is synchronised": see the option. */
int elp_verify_id_batch_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_records, uint64_t hidden_mask,
                            int with_retrieval, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags,
                            void* d_accepted);
int elp_verify_id_wire_batch_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_msgs, const void* d_msg_off,
                                 int with_retrieval, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags,
                                 void* d_accepted);
int elp_verify_id_batch_aggregated_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_records, uint64_t hidden_mask,
                                       int with_retrieval, const void* d_ad, const void* d_ad_off, size_t ad_len,
                                       const uint8_t* seed32 /* host */, void* d_flags, void* d_accepted);
int elp_ps_verify_batch_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_records, int nattr, void* d_flags,
                            void* d_accepted);
int elp_provide_id_batch_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_records, uint64_t hidden_mask,
                             const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_sigs, void* d_flags,
                             void* d_accepted);
int elp_request_id_batch_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_records, uint64_t hidden_mask,
                             const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_requests);
int elp_prove_id_batch_dev(elp_ctx* ctx, void* stream, size_t n, const void* d_records, uint64_t hidden_mask,
                           int with_retrieval, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_proofs,
                           void* d_flags, void* d_accepted);

/* bytes of device memory held by the installed key's fixed-base tables (0 without a key) */
size_t elp_key_table_bytes(const elp_ctx* ctx);

/* ---- host staging memory ------------------------------------------------------------------------------------------- */
/* Page-locked host memory for the buffers handed to the host-buffer entry points (records, messages, verdicts): copies from / to it are
 * direct DMA instead of being staged by the runtime (PCIe-inclusive rate of a 52 MB batch: ~1 ms instead of ~4 ms).  Plain malloc'ed
 * buffers stay valid inputs everywhere.  The reference has no counterpart (its data never leaves the host). */
int elp_host_alloc(elp_ctx* ctx, size_t bytes, void** out);
void elp_host_free(elp_ctx* ctx, void* p);

/* ---- measurement helpers ---------------------------------------------------------------------------------------- */
/* Times `reps` launches of the verify_id kernel with HIP events on `stream`; returns the average ms per launch. */
int elp_time_verify_id_dev(elp_ctx* ctx, void* stream, int reps, size_t n, const void* d_records, uint64_t hidden_mask,
                           int with_retrieval, const void* d_ad, const void* d_ad_off, size_t ad_len, void* d_flags,
                           void* d_accepted, float* avg_ms);
/* Base-field multiplication micro-benchmark: every lane runs `iters` dependent Montgomery products; returns ms. */
int elp_bench_fp_mul(elp_ctx* ctx, size_t lanes, int iters, float* ms);
/* Per-routine micro-benchmark (op codes in csrc/elpasso_impl.h, k_bench_op): `iters` dependent applications per lane; ms. */
int elp_bench_op(elp_ctx* ctx, int op, size_t lanes, int iters, float* ms);

#ifdef __cplusplus
}
#endif
#endif /* ELPASSO_H_ */
