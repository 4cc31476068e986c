// sweep.hip — the per-row gate / permutation / lookup quotient sweep for gfx950.
//
// Drop-in for halo2_proofs plonk::evaluation::Evaluator::evaluate_h (the part after the cosets
// exist) [UPSTREAM-RECALL src/plonk/evaluation.rs; crate pinned at /root/reference/Cargo.lock:1320-1322].
// Input is the reference's own flattened form (GraphEvaluator: constants, rotations, calculations),
// see include/zkhip.h.  Field arithmetic is exact, so any evaluation order of the same formulas
// gives identical values; the host lowers each graph to a register-machine program:
//   * Store(query) calculations become operand aliases (no copy);
//   * the closing Horner(PreviousValue, parts, Y) of the custom-gate graph becomes
//     `value = value * y + part` steps issued as soon as each part is ready, so a part's slot is
//     freed immediately (slots = live intermediates, not the number of gate polynomials);
//   * intermediates are assigned to slots by liveness.
// Device: one thread per extended row; columns are SoA cosets so lane i reads row i (32 B/lane,
// coalesced; rotations shift the whole wave's window).  Slots live in LDS laid out
// [slot][limb][lane] (conflict-free).  Custom gates, permutation and lookup terms are fused into
// one pass: h is written once (algorithmic bytes = 32 B x (distinct reads + 1)).
#include <algorithm>

#include "common.hpp"
using namespace zk;

enum : uint32_t { K_SLOT = 0, K_CONST = 1, K_COL = 2 };
enum : uint32_t { I_ADD = 0, I_SUB, I_MUL, I_SQR, I_DBL, I_NEG, I_MOV, I_MULADD, I_ACC, I_RED };
// Lazy bounds (bn254.hpp) are static per instruction — the same for every row — so the host lowering tracks
// them in sixteenths of p exactly like the el<P, B> types do, picks the k p constant of every SUB / NEG
// (field c of the instruction) and inserts a contraction (I_RED = product by one) where a sum would
// leave the representable range.  Column loads enter as 32 v (B = 32 p): the R' form of an ABI value.
static const int SW_U = 16, SW_BMAX = 120 * 16, SW_COL = 32 * 16, SW_CONST = 16;
static inline int sw_mul_bound(int a, int b) { return (a * b + 128 * SW_U - 1) / (128 * SW_U) + SW_U; }
static inline int sw_ceil_p(int b) { return (b + SW_U - 1) / SW_U; }

static inline uint32_t mk_operand(uint32_t kind, uint32_t a, uint32_t b = 0) { return (kind << 28) | ((a & 0x3fff) << 14) | (b & 0x3fff); }

struct DevIns { uint32_t op_dst, a, b, c; };

struct Section { uint32_t begin, end, result, bound; };  // instruction range, result operand and its lazy bound

// ------------------------------------------------------------------ host lowering
struct Lowering {
    std::vector<DevIns> code;
    std::vector<fe32> consts;   // R' form, canonical (raw)
    std::vector<int32_t> rots;
    std::vector<const void*> cols;  // fixed ++ advice ++ instance device pointers
    uint32_t n_fixed = 0, n_advice = 0, n_instance = 0;
    uint32_t c_zero = 0, c_one = 0, c_beta = 0, c_gamma = 0, c_theta = 0, c_y = 0, c_chal0 = 0;
    uint32_t max_slots = 0;

    uint32_t add_const(const fe32& v) { consts.push_back(v); return (uint32_t)consts.size() - 1; }
    // ABI Montgomery limbs (what the caller holds) -> table entry
    uint32_t add_const_abi(const uint64_t* p) { return add_const(fe_pack(fe_canonical<Fr>(from_abi<Fr>(mem_load(p)).v))); }
    uint32_t add_rot(int32_t r) {
        for (size_t i = 0; i < rots.size(); ++i) if (rots[i] == r) return (uint32_t)i;
        rots.push_back(r);
        return (uint32_t)rots.size() - 1;
    }
};

struct ParsedCalc { int32_t op, target; std::vector<const int32_t*> src; };

static int lower_graph(Lowering& L, const zk_graph& g, const zk_evalh_args& A, bool is_gates, Section* sec) {
    // parse
    std::vector<ParsedCalc> calcs(g.n_calculations);
    const int32_t* pc = g.code;
    const int32_t* end = g.code + g.n_code_words;
    for (uint32_t i = 0; i < g.n_calculations; ++i) {
        if (pc + 3 > end) { set_error("evaluate_h: truncated code stream"); return ZKHIP_EPROGRAM; }
        int32_t op = pc[0], target = pc[1], nsrc = pc[2];
        if (nsrc < 0 || pc + 3 + 3 * (size_t)nsrc > end) { set_error("evaluate_h: truncated calculation %u", i); return ZKHIP_EPROGRAM; }
        if (target < 0 || (uint32_t)target >= g.n_intermediates) { set_error("evaluate_h: calculation %u target out of range", i); return ZKHIP_EPROGRAM; }
        int need = (op == ZK_OP_ADD || op == ZK_OP_SUB || op == ZK_OP_MUL) ? 2 : (op == ZK_OP_HORNER ? -2 : 1);
        if (op < 0 || op > ZK_OP_STORE || (need > 0 && nsrc != need) || (need < 0 && nsrc < 2)) { set_error("evaluate_h: calculation %u malformed (op %d, nsrc %d)", i, op, nsrc); return ZKHIP_EPROGRAM; }
        calcs[i].op = op; calcs[i].target = target;
        for (int32_t s = 0; s < nsrc; ++s) calcs[i].src.push_back(pc + 3 + 3 * s);
        pc += 3 + 3 * nsrc;
    }
    const uint32_t base_const = (uint32_t)L.consts.size();
    for (uint32_t i = 0; i < g.n_constants; ++i) L.add_const_abi(g.constants + 4 * i);
    std::vector<uint32_t> rotmap(g.n_rotations);
    for (uint32_t i = 0; i < g.n_rotations; ++i) rotmap[i] = L.add_rot(g.rotations[i]);

    // operand of each intermediate: alias operand, or slot (assigned later), or undefined
    const uint32_t UNDEF = 0xffffffffu, NEEDS_SLOT = 0xfffffffeu;
    std::vector<uint32_t> val(g.n_intermediates, UNDEF);

    auto resolve = [&](const int32_t* vs, uint32_t* out) -> int {
        int32_t kind = vs[0], a = vs[1], b = vs[2];
        switch (kind) {
            case ZK_VS_CONSTANT: if (a < 0 || (uint32_t)a >= g.n_constants) goto bad; *out = mk_operand(K_CONST, base_const + a); return 0;
            case ZK_VS_INTERMEDIATE: if (a < 0 || (uint32_t)a >= g.n_intermediates || val[a] == UNDEF) goto bad; *out = val[a]; return 0;
            case ZK_VS_FIXED: if (a < 0 || (uint32_t)a >= L.n_fixed || b < 0 || (uint32_t)b >= g.n_rotations) goto bad; *out = mk_operand(K_COL, a, rotmap[b]); return 0;
            case ZK_VS_ADVICE: if (a < 0 || (uint32_t)a >= L.n_advice || b < 0 || (uint32_t)b >= g.n_rotations) goto bad; *out = mk_operand(K_COL, L.n_fixed + a, rotmap[b]); return 0;
            case ZK_VS_INSTANCE: if (a < 0 || (uint32_t)a >= L.n_instance || b < 0 || (uint32_t)b >= g.n_rotations) goto bad; *out = mk_operand(K_COL, L.n_fixed + L.n_advice + a, rotmap[b]); return 0;
            case ZK_VS_CHALLENGE: if (a < 0 || (uint32_t)a >= A.n_challenges) goto bad; *out = mk_operand(K_CONST, L.c_chal0 + a); return 0;
            case ZK_VS_BETA: *out = mk_operand(K_CONST, L.c_beta); return 0;
            case ZK_VS_GAMMA: *out = mk_operand(K_CONST, L.c_gamma); return 0;
            case ZK_VS_THETA: *out = mk_operand(K_CONST, L.c_theta); return 0;
            case ZK_VS_Y: *out = mk_operand(K_CONST, L.c_y); return 0;
            case ZK_VS_PREVIOUS: *out = mk_operand(K_CONST, L.c_zero); return 0;  // evaluate_h always starts from 0
        }
    bad:
        set_error("evaluate_h: bad value source {%d,%d,%d}", kind, a, b);
        return ZKHIP_EPROGRAM;
    };

    // Virtual instruction list; virtual registers = intermediate ids, temporaries >= n_intermediates.
    struct VIns { uint32_t op; int32_t dst; uint32_t a, b, c; int32_t ra, rb, rc; };  // r* = virtual reg read or -1
    std::vector<VIns> v;
    std::vector<int> rbound(g.n_intermediates, 0);   // lazy bound of every virtual register (sixteenths of p)
    int vbound = SW_CONST;                           // bound of the running `value` (acc mode): starts at 0
    auto opnd = [&](const int32_t* vs, uint32_t* o, int32_t* r) -> int {
        ZK_TRY(resolve(vs, o));
        *r = (*o == NEEDS_SLOT) ? vs[1] : -1;  // a real register (slot assigned below) or an alias operand
        return 0;
    };
    auto bound_of = [&](uint32_t o, int32_t r) -> int { return r >= 0 ? rbound[r] : ((o >> 28) == K_COL ? SW_COL : SW_CONST); };
    // contract operand (o, r) into a fresh temporary register; returns the new (o, r)
    auto contract = [&](uint32_t& o, int32_t& r) {
        int32_t t = (int32_t)rbound.size();
        rbound.push_back(sw_mul_bound(bound_of(o, r), SW_CONST));
        v.push_back({I_RED, t, o, 0, 0, r, -1, -1});
        o = NEEDS_SLOT; r = t;
    };
    auto emit_add = [&](int32_t dst, uint32_t oa, int32_t ra, uint32_t ob, int32_t rb) {
        while (bound_of(oa, ra) + bound_of(ob, rb) > SW_BMAX) {
            if (bound_of(oa, ra) >= bound_of(ob, rb)) contract(oa, ra); else contract(ob, rb);
        }
        v.push_back({I_ADD, dst, oa, ob, 0, ra, rb, -1});
        rbound[dst] = bound_of(oa, ra) + bound_of(ob, rb);
    };
    auto emit_sub = [&](int32_t dst, uint32_t oa, int32_t ra, uint32_t ob, int32_t rb) {
        while (bound_of(oa, ra) + (sw_ceil_p(bound_of(ob, rb)) + 1) * SW_U > SW_BMAX) {
            if (bound_of(ob, rb) > 2 * SW_U) contract(ob, rb); else contract(oa, ra);
        }
        uint32_t k = (uint32_t)sw_ceil_p(bound_of(ob, rb)) + 1;
        v.push_back({I_SUB, dst, oa, ob, k, ra, rb, -1});
        rbound[dst] = bound_of(oa, ra) + (int)k * SW_U;
    };
    // which intermediate ids are real registers
    std::vector<char> is_reg(g.n_intermediates, 0);
    const bool acc_mode = is_gates && !calcs.empty() && calcs.back().op == ZK_OP_HORNER && calcs.back().src[0][0] == ZK_VS_PREVIOUS &&
                          calcs.back().src[1][0] == ZK_VS_Y;
    std::vector<const int32_t*> acc_parts;
    if (acc_mode) for (size_t s = 2; s < calcs.back().src.size(); ++s) acc_parts.push_back(calcs.back().src[s]);
    size_t next_part = 0;
    auto flush_parts = [&]() -> int {
        while (next_part < acc_parts.size()) {
            const int32_t* vs = acc_parts[next_part];
            if (vs[0] == ZK_VS_INTERMEDIATE && (vs[1] < 0 || (uint32_t)vs[1] >= g.n_intermediates || val[vs[1]] == UNDEF)) break;
            uint32_t o; int32_t r;
            ZK_TRY(opnd(vs, &o, &r));
            while (sw_mul_bound(vbound, SW_CONST) + bound_of(o, r) > SW_BMAX) contract(o, r);
            v.push_back({I_ACC, -1, o, 0, 0, r, -1, -1});
            vbound = sw_mul_bound(vbound, SW_CONST) + bound_of(o, r);
            ++next_part;
        }
        return 0;
    };
    size_t ncalc = acc_mode ? calcs.size() - 1 : calcs.size();
    if (acc_mode) ZK_TRY(flush_parts());
    for (size_t i = 0; i < ncalc; ++i) {
        ParsedCalc& c = calcs[i];
        uint32_t o[2]; int32_t r[2];
        switch (c.op) {
            case ZK_OP_STORE: {
                ZK_TRY(opnd(c.src[0], &o[0], &r[0]));
                if (r[0] >= 0) {  // copy of a register: keep liveness simple, emit a move
                    v.push_back({I_MOV, c.target, o[0], 0, 0, r[0], -1, -1});
                    rbound[c.target] = rbound[r[0]];
                    val[c.target] = NEEDS_SLOT; is_reg[c.target] = 1;
                } else {
                    val[c.target] = o[0];  // alias: no instruction
                }
                break; }
            case ZK_OP_ADD: case ZK_OP_SUB: {
                ZK_TRY(opnd(c.src[0], &o[0], &r[0])); ZK_TRY(opnd(c.src[1], &o[1], &r[1]));
                if (c.op == ZK_OP_ADD) emit_add(c.target, o[0], r[0], o[1], r[1]); else emit_sub(c.target, o[0], r[0], o[1], r[1]);
                val[c.target] = NEEDS_SLOT; is_reg[c.target] = 1;
                break; }
            case ZK_OP_MUL: {
                ZK_TRY(opnd(c.src[0], &o[0], &r[0])); ZK_TRY(opnd(c.src[1], &o[1], &r[1]));
                v.push_back({I_MUL, c.target, o[0], o[1], 0, r[0], r[1], -1});
                rbound[c.target] = sw_mul_bound(bound_of(o[0], r[0]), bound_of(o[1], r[1]));
                val[c.target] = NEEDS_SLOT; is_reg[c.target] = 1;
                break; }
            case ZK_OP_SQUARE: {
                ZK_TRY(opnd(c.src[0], &o[0], &r[0]));
                v.push_back({I_SQR, c.target, o[0], 0, 0, r[0], -1, -1});
                rbound[c.target] = sw_mul_bound(bound_of(o[0], r[0]), bound_of(o[0], r[0]));
                val[c.target] = NEEDS_SLOT; is_reg[c.target] = 1;
                break; }
            case ZK_OP_DOUBLE: {
                ZK_TRY(opnd(c.src[0], &o[0], &r[0]));
                emit_add(c.target, o[0], r[0], o[0], r[0]);
                val[c.target] = NEEDS_SLOT; is_reg[c.target] = 1;
                break; }
            case ZK_OP_NEGATE: {
                ZK_TRY(opnd(c.src[0], &o[0], &r[0]));
                emit_sub(c.target, mk_operand(K_CONST, L.c_zero), -1, o[0], r[0]);
                val[c.target] = NEEDS_SLOT; is_reg[c.target] = 1;
                break; }
            case ZK_OP_HORNER: {
                uint32_t os, of; int32_t rs, rf;
                ZK_TRY(opnd(c.src[0], &os, &rs)); ZK_TRY(opnd(c.src[1], &of, &rf));
                v.push_back({I_MOV, c.target, os, 0, 0, rs, -1, -1});
                rbound[c.target] = bound_of(os, rs);
                for (size_t p = 2; p < c.src.size(); ++p) {
                    uint32_t op_; int32_t rp;
                    ZK_TRY(opnd(c.src[p], &op_, &rp));
                    while (sw_mul_bound(rbound[c.target], bound_of(of, rf)) + bound_of(op_, rp) > SW_BMAX) contract(op_, rp);
                    // dst = dst * factor + part   (a = dst register itself)
                    v.push_back({I_MULADD, c.target, 0xf0000000u, of, op_, c.target, rf, rp});
                    rbound[c.target] = sw_mul_bound(rbound[c.target], bound_of(of, rf)) + bound_of(op_, rp);
                }
                val[c.target] = NEEDS_SLOT; is_reg[c.target] = 1;
                break; }
        }
        if (acc_mode) ZK_TRY(flush_parts());
    }
    if (acc_mode && next_part != acc_parts.size()) { set_error("evaluate_h: Horner part %zu is never computed", next_part); return ZKHIP_EPROGRAM; }
    // result operand
    uint32_t result_op = mk_operand(K_CONST, L.c_zero);
    int32_t result_reg = -1;
    if (!acc_mode && !calcs.empty()) {
        int32_t t = calcs.back().target;
        if (val[t] == NEEDS_SLOT) result_reg = t; else result_op = val[t];
    }
    // liveness -> slots
    const size_t nreg = rbound.size();
    std::vector<int32_t> last_use(nreg, -1);
    for (size_t i = 0; i < v.size(); ++i) {
        for (int32_t r : {v[i].ra, v[i].rb, v[i].rc}) if (r >= 0) last_use[r] = (int32_t)i;
        if (v[i].dst >= 0 && last_use[v[i].dst] < (int32_t)i) last_use[v[i].dst] = (int32_t)i;
    }
    if (result_reg >= 0) last_use[result_reg] = (int32_t)v.size();
    std::vector<int32_t> slot_of(nreg, -1);
    std::vector<uint32_t> free_slots;
    uint32_t nslots = 0;
    sec->begin = (uint32_t)L.code.size();
    auto reg_operand = [&](uint32_t o, int32_t r) -> uint32_t { return r >= 0 ? mk_operand(K_SLOT, (uint32_t)slot_of[r]) : o; };
    for (size_t i = 0; i < v.size(); ++i) {
        VIns& x = v[i];
        uint32_t a = reg_operand(x.a, x.ra), b = reg_operand(x.b, x.rb), c = reg_operand(x.c, x.rc);
        // free sources whose last use is here (before allocating dst so it can be reused: reads happen before the write)
        for (int32_t r : {x.ra, x.rb, x.rc})
            if (r >= 0 && last_use[r] == (int32_t)i && slot_of[r] >= 0 && r != x.dst) { free_slots.push_back((uint32_t)slot_of[r]); slot_of[r] = -2; }
        uint32_t dst = 0;
        if (x.dst >= 0) {
            if (slot_of[x.dst] < 0) {
                if (!free_slots.empty()) { slot_of[x.dst] = (int32_t)free_slots.back(); free_slots.pop_back(); }
                else slot_of[x.dst] = (int32_t)nslots++;
            }
            dst = (uint32_t)slot_of[x.dst];
            if (last_use[x.dst] == (int32_t)i) { free_slots.push_back(dst); slot_of[x.dst] = -2; }  // dead store
        }
        if (x.op == I_MULADD) a = mk_operand(K_SLOT, dst);
        if (x.op == I_SUB) c = x.c;   // the k of the (k p) constant
        L.code.push_back({x.op | (dst << 8), a, b, c});
    }
    sec->end = (uint32_t)L.code.size();
    sec->result = acc_mode ? 0xffffffffu : result_reg >= 0 ? mk_operand(K_SLOT, (uint32_t)slot_of[result_reg]) : result_op;
    sec->bound = acc_mode ? (uint32_t)vbound : result_reg >= 0 ? (uint32_t)rbound[result_reg] : (uint32_t)bound_of(result_op, -1);
    L.max_slots = std::max(L.max_slots, nslots);
    if (nslots > 0x3fff) { set_error("evaluate_h: too many live intermediates"); return ZKHIP_EPROGRAM; }
    return ZKHIP_OK;
}

// ------------------------------------------------------------------ device
struct SweepParams {
    const DevIns* code;
    const uint32_t* consts;        // raw R'-form table
    const uint32_t* kp;            // kp[k][9]: borrow-spread limbs of k p, k <= 120
    const int32_t* rots;
    const uint32_t* const* cols;   // fixed ++ advice ++ instance (ABI form)
    uint32_t* out;
    uint32_t isize_mask, rot_scale, nslots, final_reduce;
    // Row addressing: index(row, d) = (row & hi_mask) | ((row + d) & isize_mask).  Extended domain: hi_mask = 0, isize_mask =
    // 2^extended_k - 1, rot_scale = 2^(extended_k - k).  Coset blocks (SweepCosets): hi_mask = ~(n - 1) keeps the block, isize_mask =
    // n - 1, rot_scale = 1; block r = row >> coset_shift selects the coset's constants (coset_shift = 31 otherwise: always 0).
    uint32_t hi_mask, coset_shift;
    uint32_t row0;                 // first row of this launch (row-range entry point: out[0] is row0's value)
    Section gates;
    // permutation
    uint32_t n_perm_sets, n_perm_cols, chunk_len; int32_t last_rot;
    const uint32_t* perm_col_slot;        // index into cols[]
    const uint32_t* const* sigma;         // n_perm_cols
    const uint32_t* const* perm_z;        // n_perm_sets
    const uint32_t* l0; const uint32_t* l_last; const uint32_t* l_active;
    const uint32_t* xt_lo; const uint32_t* xt_hi; uint32_t xt_h;  // extended_omega^i = lo[i mod 2^h] * hi[i >> h] (raw R' form)
    uint32_t c_beta, c_gamma, c_y, c_one, c_delta, c_delta_start;
    // lookups
    uint32_t n_lookups;
    const Section* lookup_secs;
    const uint32_t* const* lookup_z; const uint32_t* const* lookup_a; const uint32_t* const* lookup_s;
};

extern __shared__ uint32_t sweep_lds[];   // [slot][9][lane]

__device__ __forceinline__ fe slot_read(uint32_t slot) {
    fe r;
    const uint32_t* p = sweep_lds + (size_t)slot * 9 * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p[i * blockDim.x];
    return r;
}
__device__ __forceinline__ void slot_write(uint32_t slot, const fe& v) {
    uint32_t* p = sweep_lds + (size_t)slot * 9 * blockDim.x + threadIdx.x;
#pragma unroll
    for (int i = 0; i < 9; ++i) p[i * blockDim.x] = v.l[i];
}
// operand -> limbs.  Bounds are the host's business (see the lowering); columns enter as 32 v.
__device__ __forceinline__ fe fetch(const SweepParams& P, uint32_t o, uint32_t row) {
    uint32_t kind = o >> 28, a = (o >> 14) & 0x3fff, b = o & 0x3fff;
    if (kind == K_SLOT) return slot_read(a);
    if (kind == K_CONST) return fe_split<0>(mem_load(P.consts + (size_t)a * 8));
    uint32_t r = (row & P.hi_mask) | ((uint32_t)((int32_t)row + P.rots[b] * (int32_t)P.rot_scale) & P.isize_mask);
    return fe_split<5>(mem_load(P.cols[a] + (size_t)r * 8));
}
__device__ __forceinline__ fe raw_add(const fe& a, const fe& b) {
    fe r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
    fe_normalize(r);
    return r;
}
__device__ __forceinline__ fe raw_sub(const SweepParams& P, const fe& a, const fe& b, uint32_t k) {
    fe r;
    const uint32_t* kp = P.kp + (size_t)k * 9;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + kp[i] - b.l[i];
    fe_normalize(r);
    return r;
}
__device__ __forceinline__ fe raw_one() {
    fe o;
#pragma unroll
    for (int i = 0; i < 9; ++i) o.l[i] = Fr::ONE[i];
    return o;
}
__device__ __forceinline__ void run_section(const SweepParams& P, const Section& s, uint32_t row, fe& value, const fe& y) {
    for (uint32_t pc = s.begin; pc < s.end; ++pc) {
        DevIns in = P.code[pc];
        uint32_t op = in.op_dst & 0xff, dst = in.op_dst >> 8;
        fe a = fetch(P, in.a, row);
        fe r;
        switch (op) {
            case I_ADD: r = raw_add(a, fetch(P, in.b, row)); break;
            case I_SUB: r = raw_sub(P, a, fetch(P, in.b, row), in.c); break;
            case I_MUL: r = fe_mul_raw<Fr>(a, fetch(P, in.b, row)); break;
            case I_SQR: r = fe_mul_raw<Fr>(a, a); break;
            case I_MOV: r = a; break;
            case I_RED: r = fe_mul_raw<Fr>(a, raw_one()); break;
            case I_MULADD: r = raw_add(fe_mul_raw<Fr>(a, fetch(P, in.b, row)), fetch(P, in.c, row)); break;
            default: /* I_ACC */ value = raw_add(fe_mul_raw<Fr>(value, y), a); continue;
        }
        slot_write(dst, r);
    }
}

// The fixed permutation / lookup terms are written with the typed elements, so their bounds are checked
// at compile time; `value` and a lookup's table_value arrive from the interpreter with host-checked
// bounds <= 120 p, which is what their types say.
using elv = el<Fr, BMAX>;
using elc = el<Fr, 32 * U>;   // a column value as loaded
__device__ __forceinline__ elc ldc(const uint32_t* col, uint32_t row) { return load_x32<Fr>(col + (size_t)row * 8); }
__device__ __forceinline__ el1<Fr> ldk(const SweepParams& P, uint32_t idx) { return load_raw<Fr>(P.consts + (size_t)idx * 8); }

__global__ void __launch_bounds__(128) k_sweep(SweepParams P) {
    uint32_t row = P.row0 + blockIdx.x * blockDim.x + threadIdx.x;  // the row count is a multiple of the block size
    const el1<Fr> y = ldk(P, P.c_y);
    elv value(fe_zero());
    run_section(P, P.gates, row, value.v, y.v);
    if (P.gates.result != 0xffffffffu) value = elv(fetch(P, P.gates.result, row));  // 0xffffffff: `value` already holds it

    const uint32_t mask = P.isize_mask, hi = row & P.hi_mask;
    if (P.n_perm_sets) {
        const el1<Fr> beta = ldk(P, P.c_beta), gamma = ldk(P, P.c_gamma), k_one = ldk(P, P.c_one);
        uint32_t r_next = hi | ((row + P.rot_scale) & mask);
        uint32_t r_last = hi | ((uint32_t)((int32_t)row + P.last_rot * (int32_t)P.rot_scale) & mask);
        elc l0 = ldc(P.l0, row), ll = ldc(P.l_last, row), la = ldc(P.l_active, row);
        elc zf = ldc(P.perm_z[0], row);
        value = muladd2(value, y, k_one - zf, l0);   // value y + (1 - z) l0: two products, one reduction
        elc zl = ldc(P.perm_z[P.n_perm_sets - 1], row);
        value = muladd2(value, y, sqr(zl) - zl, ll);
        for (uint32_t s = 1; s < P.n_perm_sets; ++s) {
            elc zi = ldc(P.perm_z[s], row), zp = ldc(P.perm_z[s - 1], r_last);
            value = muladd2(value, y, zi - zp, l0);
        }
        // current_delta = beta * g_coset * extended_omega^row   (coset blocks: beta * s_r * omega^(row mod n), s_r from the table)
        const uint32_t xi = row & mask;
        el2<Fr> xw = load_raw<Fr>(P.xt_lo + (size_t)(xi & ((1u << P.xt_h) - 1)) * 8) * load_raw<Fr>(P.xt_hi + (size_t)(xi >> P.xt_h) * 8);
        el2<Fr> cur = ldk(P, P.c_delta_start + (row >> P.coset_shift)) * xw;
        const el1<Fr> delta = ldk(P, P.c_delta);
        for (uint32_t s = 0; s < P.n_perm_sets; ++s) {
            uint32_t c0 = s * P.chunk_len, c1 = min(c0 + P.chunk_len, P.n_perm_cols);
            elc left = ldc(P.perm_z[s], r_next), right = ldc(P.perm_z[s], row);
            for (uint32_t c = c0; c < c1; ++c) {
                elc v = ldc(P.cols[P.perm_col_slot[c]], row);
                elc sg = ldc(P.sigma[c], row);
                left = left * (v + beta * sg + gamma);
                right = right * (v + cur + gamma);
                cur = cur * delta;
            }
            value = muladd2(value, y, left - right, la);
        }
    }
    if (P.n_lookups) {
        const el1<Fr> beta = ldk(P, P.c_beta), gamma = ldk(P, P.c_gamma), k_one = ldk(P, P.c_one);
        elc l0 = ldc(P.l0, row), ll = ldc(P.l_last, row), la = ldc(P.l_active, row);
        uint32_t r_next = hi | ((row + P.rot_scale) & mask), r_prev = hi | ((row - P.rot_scale) & mask);
        for (uint32_t n = 0; n < P.n_lookups; ++n) {
            Section sec = P.lookup_secs[n];
            fe dummy = fe_zero();
            run_section(P, sec, row, dummy, y.v);
            elv table_value(fetch(P, sec.result, row));
            elc z = ldc(P.lookup_z[n], row), zn = ldc(P.lookup_z[n], r_next);
            elc a = ldc(P.lookup_a[n], row), ap = ldc(P.lookup_a[n], r_prev);
            elc sv = ldc(P.lookup_s[n], row);
            auto a_minus_s = a - sv;
            value = muladd2(value, y, k_one - z, l0);
            value = muladd2(value, y, sqr(z) - z, ll);
            auto t = (a + beta) * (sv + gamma) * zn - z * table_value;
            value = muladd2(value, y, t, la);
            value = muladd2(value, y, a_minus_s, l0);
            value = muladd2(value, y, a_minus_s * (a - ap), la);
        }
    }
    // back to the ABI form; the host sets final_reduce when the gate program alone could leave > 88 p
    if (P.final_reduce) value = reduce(value);
    store_div32<Fr>(P.out + (size_t)(row - P.row0) * 8, el<Fr, 88 * U>(value.v));
}

// ------------------------------------------------------------------ entry point
static int evaluate_h_rows(zkhip_ctx* ctx, const zk_evalh_args* A, size_t first_row, size_t n_rows, void* d_out, const zk::SweepCosets* cs = nullptr);
int zk::evaluate_h_cosets(zkhip_ctx* ctx, const zk_evalh_args* A, const SweepCosets* cs, size_t first_row, size_t n_rows, void* d_out) {
    if (!cs || !cs->q || !cs->shifts_abi || !cs->omega_abi) { set_error("evaluate_h_cosets: null argument"); return ZKHIP_EINVAL; }
    return evaluate_h_rows(ctx, A, first_row, n_rows, d_out, cs);
}
extern "C" int zkhip_evaluate_h_device(zkhip_ctx* ctx, const zk_evalh_args* A, void* d_out) {
    if (!A) { set_error("zkhip_evaluate_h_device: null argument"); return ZKHIP_EINVAL; }
    return evaluate_h_rows(ctx, A, 0, (size_t)1 << (A->extended_k <= 26 ? A->extended_k : 0), d_out);
}
extern "C" int zkhip_evaluate_h_rows_device(zkhip_ctx* ctx, const zk_evalh_args* A, size_t first_row, size_t n_rows, void* d_out) {
    return evaluate_h_rows(ctx, A, first_row, n_rows, d_out);
}
static int evaluate_h_rows(zkhip_ctx* ctx, const zk_evalh_args* A, size_t first_row, size_t n_rows, void* d_out, const zk::SweepCosets* cs) {
    if (!ctx || !A || !d_out) { set_error("zkhip_evaluate_h_device: null argument"); return ZKHIP_EINVAL; }
    if (A->extended_k < A->k || A->extended_k > 26 || A->extended_k < 2) { set_error("zkhip_evaluate_h_device: extended_k = %u unsupported (2..26)", A->extended_k); return ZKHIP_EINVAL; }
    if (A->n_perm_sets && A->cs_degree < 3) { set_error("zkhip_evaluate_h_device: cs_degree < 3 with a permutation argument"); return ZKHIP_EINVAL; }
    if (A->n_fixed + A->n_advice + A->n_instance > 0x3fff) { set_error("zkhip_evaluate_h_device: too many columns"); return ZKHIP_EINVAL; }
    Lowering L;
    L.n_fixed = A->n_fixed; L.n_advice = A->n_advice; L.n_instance = A->n_instance;
    for (uint32_t i = 0; i < A->n_fixed; ++i) L.cols.push_back(A->fixed_cosets[i]);
    for (uint32_t i = 0; i < A->n_advice; ++i) L.cols.push_back(A->advice_cosets[i]);
    for (uint32_t i = 0; i < A->n_instance; ++i) L.cols.push_back(A->instance_cosets[i]);
    auto cst = [&](const uint64_t* p) { return L.add_const_abi(p); };
    L.c_zero = L.add_const(fe_pack(fe_zero()));
    L.c_one = L.add_const(fe_pack(fe_canonical<Fr>(one<Fr>().v)));
    L.c_beta = cst(A->beta); L.c_gamma = cst(A->gamma); L.c_theta = cst(A->theta); L.c_y = cst(A->y);
    uint32_t c_delta = cst(A->delta);
    // beta * (the coset generator): one entry, or one per coset block (consecutive)
    uint32_t c_delta_start = 0;
    for (uint32_t r = 0; r < (cs ? cs->q : 1u); ++r) {
        const uint64_t* shift = cs ? cs->shifts_abi + 4 * r : A->g_coset;
        fe32 sh;   // the coset table may be 8-byte aligned only
        memcpy(sh.w, shift, 32);
        uint32_t idx = L.add_const(fe_pack(fe_canonical<Fr>((from_abi<Fr>(mem_load(A->beta)) * from_abi<Fr>(sh)).v)));
        if (r == 0) c_delta_start = idx;
    }
    L.c_chal0 = (uint32_t)L.consts.size();
    for (uint32_t i = 0; i < A->n_challenges; ++i) cst(A->challenges + 4 * i);

    Section gates;
    ZK_TRY(lower_graph(L, A->custom_gates, *A, true, &gates));
    std::vector<Section> lsecs(A->n_lookups);
    for (uint32_t i = 0; i < A->n_lookups; ++i) ZK_TRY(lower_graph(L, A->lookup_graphs[i], *A, false, &lsecs[i]));
    if (L.consts.size() > 0x3fff || L.rots.size() > 0x3fff) { set_error("zkhip_evaluate_h_device: program tables too large"); return ZKHIP_EPROGRAM; }

    // pack everything into one upload
    hipStream_t st = ctx->stream;
    // rows of the sweep: the extended domain, or q blocks of n
    const size_t isize = cs ? (size_t)cs->q << A->k : (size_t)1 << A->extended_k;
    std::vector<uint32_t> perm_slot(A->n_perm_columns);
    for (uint32_t c = 0; c < A->n_perm_columns; ++c) {
        uint32_t ty = A->perm_column_type[c], ix = A->perm_column_index[c];
        uint32_t lim = ty == 0 ? A->n_advice : ty == 1 ? A->n_fixed : ty == 2 ? A->n_instance : 0;
        if (ix >= lim) { set_error("zkhip_evaluate_h_device: permutation column %u out of range", c); return ZKHIP_EINVAL; }
        perm_slot[c] = ty == 0 ? A->n_fixed + ix : ty == 1 ? ix : A->n_fixed + A->n_advice + ix;
    }
    std::vector<char> blob;
    auto put = [&](const void* p, size_t bytes) { size_t off = (blob.size() + 31) & ~(size_t)31; blob.resize(off + bytes); if (bytes) memcpy(blob.data() + off, p, bytes); return off; };
    size_t o_code = put(L.code.data(), L.code.size() * sizeof(DevIns));
    size_t o_consts = put(L.consts.data(), L.consts.size() * 32);
    std::vector<uint32_t> kp(121 * 9, 0);
    for (uint32_t k = 1; k <= 120; ++k)
        for (int i = 0; i < 9; ++i) kp[k * 9 + i] = kp_spread<Fr>(k, i);
    size_t o_kp = put(kp.data(), kp.size() * 4);
    size_t o_rots = put(L.rots.data(), L.rots.size() * 4);
    size_t o_cols = put(L.cols.data(), L.cols.size() * sizeof(void*));
    size_t o_pslot = put(perm_slot.data(), perm_slot.size() * 4);
    size_t o_sigma = put(A->perm_sigma_cosets, A->n_perm_columns * sizeof(void*));
    size_t o_pz = put(A->perm_product_cosets, A->n_perm_sets * sizeof(void*));
    size_t o_lsec = put(lsecs.data(), lsecs.size() * sizeof(Section));
    size_t o_lz = put(A->lookup_product_cosets, A->n_lookups * sizeof(void*));
    size_t o_la = put(A->lookup_input_cosets, A->n_lookups * sizeof(void*));
    size_t o_ls = put(A->lookup_table_cosets, A->n_lookups * sizeof(void*));
    put(nullptr, 0);
    void* d_blob;
    ZK_TRY(ctx->get_scratch("sweep_blob", blob.size() + 64, &d_blob));
    ZK_TRY(ctx->upload(d_blob, blob.data(), blob.size()));   // staged: no stream synchronisation for the host temporary

    SweepParams P;
    memset(&P, 0, sizeof P);
    char* b = (char*)d_blob;
    P.code = (const DevIns*)(b + o_code);
    P.consts = (const uint32_t*)(b + o_consts);
    P.kp = (const uint32_t*)(b + o_kp);
    P.final_reduce = (A->n_perm_sets == 0 && A->n_lookups == 0 && gates.bound > 88 * 16) ? 1u : 0u;
    P.rots = (const int32_t*)(b + o_rots);
    P.cols = (const uint32_t* const*)(b + o_cols);
    P.out = (uint32_t*)d_out;
    P.isize_mask = cs ? (1u << A->k) - 1 : (uint32_t)isize - 1;
    P.hi_mask = cs ? ~P.isize_mask : 0u;
    P.coset_shift = cs ? A->k : 31u;
    P.row0 = (uint32_t)first_row;
    P.rot_scale = cs ? 1u : 1u << (A->extended_k - A->k);
    P.nslots = L.max_slots;
    P.gates = gates;
    P.n_perm_sets = A->n_perm_sets; P.n_perm_cols = A->n_perm_columns;
    P.chunk_len = A->cs_degree >= 3 ? A->cs_degree - 2 : 1;
    P.last_rot = -((int32_t)A->blinding_factors + 1);
    P.perm_col_slot = (const uint32_t*)(b + o_pslot);
    P.sigma = (const uint32_t* const*)(b + o_sigma);
    P.perm_z = (const uint32_t* const*)(b + o_pz);
    P.l0 = (const uint32_t*)A->l0; P.l_last = (const uint32_t*)A->l_last; P.l_active = (const uint32_t*)A->l_active_row;
    P.c_beta = L.c_beta; P.c_gamma = L.c_gamma; P.c_y = L.c_y; P.c_one = L.c_one; P.c_delta = c_delta; P.c_delta_start = c_delta_start;
    P.n_lookups = A->n_lookups;
    P.lookup_secs = (const Section*)(b + o_lsec);
    P.lookup_z = (const uint32_t* const*)(b + o_lz);
    P.lookup_a = (const uint32_t* const*)(b + o_la);
    P.lookup_s = (const uint32_t* const*)(b + o_ls);
    if (A->n_perm_sets) {
        if (!A->l0 || !A->l_last || !A->l_active_row) { set_error("zkhip_evaluate_h_device: l0/l_last/l_active_row missing"); return ZKHIP_EINVAL; }
        const zkhip_ctx::Twiddle* xt;
        ZK_TRY(cs ? ctx->get_twiddles(cs->omega_abi, A->k, &xt) : ctx->get_twiddles(A->extended_omega, A->extended_k, &xt));
        P.xt_lo = (const uint32_t*)xt->d_lo;
        P.xt_hi = (const uint32_t*)xt->d_hi;
        P.xt_h = xt->h;
    }
    // one thread per row; domains below 64 rows (k = 4, 5 circuits: the lookup compression runs on 2^k rows) take one partial wave
    // (q blocks of 16 or 32 rows — 48, 96, 80 ... — go in half or quarter waves)
    const unsigned block = n_rows % 128 == 0 ? 128 : n_rows % 64 == 0 ? 64 : n_rows < 64 ? (unsigned)n_rows : n_rows % 32 == 0 ? 32 : 16;
    if (first_row + n_rows > isize || n_rows == 0 || (n_rows % 16 && n_rows > 64)) {
        set_error("zkhip_evaluate_h_rows_device: rows [%zu, %zu) out of the domain or not a multiple of 16", first_row, first_row + n_rows);
        return ZKHIP_EINVAL;
    }
    size_t lds = (size_t)std::max<uint32_t>(L.max_slots, 1) * 9 * 4 * block;
    if (lds > 160 * 1024) { set_error("zkhip_evaluate_h_device: %u live intermediates exceed the LDS budget", L.max_slots); return ZKHIP_EPROGRAM; }
    if (lds > 64 * 1024) ZK_HIP(hipFuncSetAttribute((const void*)k_sweep, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ProfScope ps(ctx, "sweep");
    hipLaunchKernelGGL(k_sweep, dim3((unsigned)(n_rows / block)), dim3(block), lds, st, P);
    ZK_LAUNCH_CHECK();
    return ZKHIP_OK;
}
