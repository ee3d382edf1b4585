// Host-side circuit planner: gate list -> passes of LDS-resident stages.
//
// What it does, in order:
//   1. drops identity gates and rewrites  cx(a,b) p(l,b) cx(a,b)  as one diagonal
//      term "phase l where bit a != bit b" (the ZZ-feature-map entangler,
//      reference quantum_featuremaps.py:75-78) -- same multiplications, no swaps;
//   2. merges runs of diagonal gates (p, rz, cz, the term above) into one DIAG stage,
//      letting diagonal gates sink past later non-diagonal gates they commute with;
//   1b. multiplies runs of single-qubit gates on one qubit (two or more of them non-diagonal, nothing
//      two-qubit in between) into ONE 2x2 matrix per sample (QK_UC) -- a GA genome circuit draws a
//      gate per qubit and layer and only a few cx, so most of it collapses into the initial
//      product state and a handful of general single-qubit gates;
//   3. splits the stage list into passes whose non-diagonal gates only touch
//      "local" qubits (<= 13, the part of the state one workgroup keeps in LDS).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "qk_common.h"
#include "qk_reg.h"
#include "qk_wave.h"

namespace qk {
namespace {

struct Item {
    bool diag;
    int32_t kind;   // diag: TermKind ; non-diag: StageKind (ST_U1 / ST_CX / ST_ECR)
    int32_t gate;   // qk_gate for ST_U1
    int32_t q0, q1;
    int32_t expr;   // global expr index or -1
};

struct Proto {      // a stage before virtual-bit translation
    bool diag;
    std::vector<Item> terms;  // diag
    Item nd;                  // non-diag
};

bool touches(const Item& t, int q) { return t.q0 == q || (t.q1 >= 0 && t.q1 == q); }

}  // namespace

int plan_circuit(qk_circuit& c, const qk_op* ops, int n_ops, const int32_t* expr_off, int n_expr, bool allow_narrow) {
    const int n = c.n_qubits;
    auto bad_q = [&](int q) { return q < 0 || q >= n; };

    // ---- 1. classify + peephole ------------------------------------------
    static const bool basis_pairs_off = [] { const char* e = std::getenv("QK_PLAN_BASIS_CONTROLS"); return e && std::atoi(e) == 0; }();
    std::vector<Item> items;
    // cx, cz and ecr are their own inverses: two of them on the same qubits with nothing in between on either qubit (id
    // gates aside) are dropped together -- a genome draws a qubit's partner from its neighbours, so such pairs are frequent.
    // hist[q]: the live items that touched q, in order.
    std::vector<char> dead;
    std::vector<std::vector<int>> hist(n);
    auto push = [&](const Item& it) {
        const int idx = (int)items.size();
        items.push_back(it);
        dead.push_back(0);
        hist[it.q0].push_back(idx);
        if (it.q1 >= 0) hist[it.q1].push_back(idx);
    };
    // (... and gates that commute with the pair may stand between its halves: diagonal gates on the control of a cx and on
    // either qubit of a cz, x / sx / rx on the target of a cx and on the second qubit of an ecr)
    auto cancels = [&](bool diag, int kind, int q0, int q1) {
        if (basis_pairs_off) return false;
        auto commutes = [&](const Item& it, bool first) {
            if (it.q1 >= 0) return false;
            if (it.diag) return diag || (kind == ST_CX && first);
            return !diag && !first && it.kind == ST_U1 && (it.gate == QK_X || it.gate == QK_SX || it.gate == QK_RX);
        };
        auto last_blocking = [&](int q, bool first) {
            int k = (int)hist[q].size() - 1;
            while (k >= 0 && commutes(items[(size_t)hist[q][(size_t)k]], first)) --k;
            return k;
        };
        const int k0 = last_blocking(q0, true), k1 = last_blocking(q1, false);
        if (k0 < 0 || k1 < 0 || hist[q0][(size_t)k0] != hist[q1][(size_t)k1]) return false;
        const int j = hist[q0][(size_t)k0];
        const Item& p = items[(size_t)j];
        const bool same = p.diag == diag && p.kind == kind && ((p.q0 == q0 && p.q1 == q1) || (diag && p.q0 == q1 && p.q1 == q0));
        if (!same) return false;
        dead[(size_t)j] = 1;
        hist[q0].erase(hist[q0].begin() + k0);
        hist[q1].erase(hist[q1].begin() + k1);
        return true;
    };
    // basis[q]: 0 / 1 while only id, x and diagonal gates have acted on q -- the qubit is still |0> or |1> up to a phase --
    // and -1 after anything else.  A cx whose control is still |0> is the identity and a cz with either qubit still |0> as
    // well; a cx whose control is |1> is an x on its target.  (GA genomes draw their gates at random: the first gate on a
    // qubit is often the control of a cx, and without it the qubit's single-qubit gates join the product state.)
    // An ecr is rz(pi/2) on its first qubit and sx on its second, then cx(first, second), then x on the first (exact): while
    // its first qubit is still a basis state the cx is nothing or an x, and the whole gate is single-qubit gates -- the
    // hardware-native genomes open many qubits that way.  The rz(pi/2) is a phase that must stay: its angle is one more
    // constant program at the end of the circuit's own.
    static const bool basis_off = [] { const char* e = std::getenv("QK_PLAN_BASIS_CONTROLS"); return e && std::atoi(e) == 0; }();
    std::vector<int> basis(n, basis_off ? -1 : 0);
    std::vector<int32_t> eoff(expr_off ? expr_off : nullptr, expr_off ? expr_off + n_expr + 1 : nullptr);
    if (eoff.empty()) eoff.push_back(0);
    int half_pi_expr = -1;
    auto half_pi = [&]() {
        if (half_pi_expr < 0) {
            c.consts.push_back(1.57079632679489661923);
            c.prog.push_back(qk_instr{QK_EX_CONST, (int32_t)c.consts.size() - 1});
            eoff.push_back((int32_t)c.prog.size());
            half_pi_expr = (int)eoff.size() - 2;
        }
        return half_pi_expr;
    };
    for (int i = 0; i < n_ops; ++i) {
        qk_op o = ops[i];
        const bool two = (o.gate == QK_CX || o.gate == QK_CZ || o.gate == QK_ECR);
        const bool par = (o.gate == QK_RX || o.gate == QK_RY || o.gate == QK_RZ || o.gate == QK_P);
        if (o.gate < QK_ID || o.gate > QK_ECR) return fail(QK_ERR_INVALID, "unknown gate code in op table");
        if (bad_q(o.q0) || (two && (bad_q(o.q1) || o.q1 == o.q0)))
            return fail(QK_ERR_INVALID, "gate qubit index out of range");
        if (par && (o.expr < 0 || o.expr >= n_expr))
            return fail(QK_ERR_INVALID, "parametric gate without a valid angle program");
        if (o.gate == QK_ID) continue;
        if (o.gate == QK_CX && basis[o.q0] == 0) continue;
        if (o.gate == QK_CZ && (basis[o.q0] == 0 || basis[o.q1] == 0)) continue;
        if (o.gate == QK_CX && basis[o.q0] == 1) { o.gate = QK_X; o.q0 = o.q1; o.q1 = -1; }
        if (o.gate == QK_ECR && basis[o.q0] >= 0) {
            push({true, TM_RZ, 0, o.q0, -1, half_pi()});
            push({false, ST_U1, QK_SX, o.q1, -1, -1});
            if (basis[o.q0] == 1) push({false, ST_U1, QK_X, o.q1, -1, -1});
            push({false, ST_U1, QK_X, o.q0, -1, -1});
            basis[o.q0] = 1 - basis[o.q0];
            basis[o.q1] = -1;
            continue;
        }
        if (o.gate == QK_X) basis[o.q0] = basis[o.q0] < 0 ? -1 : 1 - basis[o.q0];
        else if (o.gate == QK_CX) basis[o.q1] = -1;      // (control not a basis state: the target is entangled with it)
        else if (o.gate == QK_ECR) basis[o.q0] = basis[o.q1] = -1;
        else if (o.gate != QK_RZ && o.gate != QK_P && o.gate != QK_CZ) basis[o.q0] = -1;
        if (o.gate == QK_CX && i + 2 < n_ops) {
            const qk_op& p = ops[i + 1];
            const qk_op& x2 = ops[i + 2];
            if (p.gate == QK_P && p.q0 == o.q1 && p.expr >= 0 && p.expr < n_expr && x2.gate == QK_CX &&
                x2.q0 == o.q0 && x2.q1 == o.q1) {
                push({true, TM_ZZP, 0, o.q0, o.q1, p.expr});
                i += 2;
                continue;
            }
        }
        switch (o.gate) {
            case QK_H: case QK_X: case QK_SX: case QK_RX: case QK_RY:
                push({false, ST_U1, o.gate, o.q0, -1, par ? o.expr : -1});
                break;
            case QK_RZ: push({true, TM_RZ, 0, o.q0, -1, o.expr}); break;
            case QK_P: push({true, TM_P, 0, o.q0, -1, o.expr}); break;
            case QK_CZ: if (!cancels(true, TM_CZ, o.q0, o.q1)) push({true, TM_CZ, 0, o.q0, o.q1, -1}); break;
            case QK_CX: if (!cancels(false, ST_CX, o.q0, o.q1)) push({false, ST_CX, o.gate, o.q0, o.q1, -1}); break;
            case QK_ECR: if (!cancels(false, ST_ECR, o.q0, o.q1)) push({false, ST_ECR, o.gate, o.q0, o.q1, -1}); break;
            default: break;
        }
    }
    {      // (without the cancelled pairs)
        std::vector<Item> live;
        for (size_t k = 0; k < items.size(); ++k)
            if (!dead[k]) live.push_back(items[k]);
        items.swap(live);
    }
    c.n_gates = (int)items.size();
    expr_off = eoff.data();      // (the caller's angle programs, and the constant above when an ecr needed it)
    n_expr = (int)eoff.size() - 1;

    bool two_qubit_diag = false;      // cz or cx p cx anywhere in the circuit
    for (const Item& it : items) two_qubit_diag = two_qubit_diag || (it.diag && it.q1 >= 0);

    // ---- 1b. runs of single-qubit gates on one qubit -> one composite matrix -------------------
    // chain[q]: the single-qubit items on q since the last two-qubit item that touched q.  Gates on
    // different qubits commute, so a chain may be emitted as a block right before the next two-qubit
    // item on its qubit (or at the end).  A chain with two or more non-diagonal gates becomes one QK_UC;
    // any other chain is emitted as it was (lone h / rx / ... keep their cheap special forms, diagonal
    // gates keep merging into diagonal stages).  Circuits without such a chain keep their gate order.
    struct Factor { int32_t gate, expr; };
    std::vector<std::vector<Factor>> comps;
    {
        auto single = [](const Item& it) { return it.diag ? (it.kind == TM_P || it.kind == TM_RZ) : it.kind == ST_U1; };
        // Circuits whose diagonal gates are all single-qubit ones (GA genomes: rz / p between sx, x, ecr, cx): a chain with
        // ONE non-diagonal gate and diagonal gates around it is a composite as well -- rz sx rz is how the hardware-native
        // alphabet writes any rotation, and left apart each rz becomes a term of a diagonal stage of its own (three phase
        // tables per stage and sample, a walk over every amplitude) where one general butterfly does it.  Not where the
        // chain is the first thing on its qubit and starts with a Hadamard (h p: the Hadamards are the constant product
        // state, the phases one diagonal stage: the Z map), and not in circuits with cz / zz terms, whose stages exist anyway.
        static const bool env_off = [] { const char* e = std::getenv("QK_PLAN_RZ_CHAINS"); return e && std::atoi(e) == 0; }();
        const bool rz_chains = !two_qubit_diag && !env_off;
        std::vector<std::vector<Item>> chain(n);
        std::vector<Item> out;
        std::vector<bool> seen(n, false);      // something has acted on the qubit before the chain
        bool any = false;
        auto flush_q = [&](int q) {
            std::vector<Item>& ch = chain[q];
            int nd = 0;
            for (const Item& g : ch) nd += g.diag ? 0 : 1;
            const bool around = rz_chains && nd == 1 && ch.size() >= 2 && !(!seen[q] && !ch[0].diag && ch[0].gate == QK_H);
            if (!ch.empty()) seen[q] = true;
            if (nd < 2 && !around) {
                out.insert(out.end(), ch.begin(), ch.end());
            } else {
                // the whole chain, diagonal gates at its head and tail included: a composite that is the first
                // thing on its qubit joins the initial product state, and no diagonal stage is left behind
                any = true;
                std::vector<Factor> fs;
                for (const Item& g : ch)
                    fs.push_back(Factor{g.diag ? (g.kind == TM_P ? (int32_t)QK_P : (int32_t)QK_RZ) : g.gate, g.expr});
                out.push_back(Item{false, ST_U1, QK_UC, q, -1, (int32_t)comps.size()});
                comps.push_back(std::move(fs));
            }
            ch.clear();
        };
        for (const Item& it : items) {
            if (single(it)) { chain[it.q0].push_back(it); continue; }
            flush_q(it.q0);
            if (it.q1 >= 0) flush_q(it.q1);
            seen[it.q0] = true;
            if (it.q1 >= 0) seen[it.q1] = true;
            out.push_back(it);
        }
        for (int q = 0; q < n; ++q) flush_q(q);
        if (any) items.swap(out);
        else comps.clear();
    }

    // A single-qubit gate that is the very first operation on its qubit acts on |0> and commutes
    // with everything before it: all of those are hoisted to the start of the circuit, where they
    // turn |0...0> into a product state the first pass writes directly (ST_INIT) -- no sweeps and
    // no need for the qubit to be local.
    std::vector<Item> inits;
    {
        std::vector<Item> rest;
        uint32_t touched = 0;
        for (const Item& it : items) {
            if (!it.diag && it.kind == ST_U1 && !((touched >> it.q0) & 1u)) inits.push_back(it);
            else rest.push_back(it);
            touched |= 1u << it.q0;
            if (it.q1 >= 0) touched |= 1u << it.q1;
        }
        items.swap(rest);
    }

    // ---- 2. merge diagonal runs, sinking them past commuting gates ---------
    std::vector<Proto> seq;
    std::vector<Item> pending;
    auto flush = [&]() {
        if (pending.empty()) return;
        Proto p;
        p.diag = true;
        p.terms.swap(pending);
        p.nd = Item{};
        seq.push_back(std::move(p));
    };
    for (const Item& it : items) {
        if (it.diag) {
            pending.push_back(it);
            if ((int)pending.size() >= kMaxTermPerPass - 64) flush();  // a stage's terms must fit one pass
            continue;
        }
        // qubits on which this gate is NOT diagonal (a cx is diagonal on its control)
        int nq[2] = {it.kind == ST_CX ? it.q1 : it.q0, it.kind == ST_ECR ? it.q1 : -1};
        bool blocked = false;
        for (const Item& t : pending)
            for (int q : nq)
                if (q >= 0 && touches(t, q)) blocked = true;
        if (blocked) flush();
        Proto p;
        p.diag = false;
        p.nd = it;
        seq.push_back(std::move(p));
    }
    flush();

    {
        int n_u1 = 0, n_uc = 0, n_cx = 0, n_ecr = 0, n_diag = 0, n_terms = 0, n_fact = 0;
        for (const Proto& p : seq) {
            if (p.diag) { ++n_diag; n_terms += (int)p.terms.size(); }
            else if (p.nd.kind == ST_U1) { if (p.nd.gate == QK_UC) ++n_uc; else ++n_u1; }
            else if (p.nd.kind == ST_CX) ++n_cx;
            else ++n_ecr;
        }
        for (const auto& f : comps) n_fact += (int)f.size();
        int init_uc = 0;
        for (const Item& g : inits) init_uc += g.gate == QK_UC;
        c.plan_summary = "gates " + std::to_string(c.n_gates) + ": product-state gates " + std::to_string(inits.size()) +
                         " (" + std::to_string(init_uc) + " composite), then u1 " + std::to_string(n_u1) + ", composite u1 " +
                         std::to_string(n_uc) + ", cx " + std::to_string(n_cx) + ", ecr " + std::to_string(n_ecr) +
                         ", diagonal stages " + std::to_string(n_diag) + " with " + std::to_string(n_terms) + " terms; composites " +
                         std::to_string(comps.size()) + " over " + std::to_string(n_fact) + " factors";
    }

    // ---- 2b. the wave-resident program (up to 10 qubits: the state lives in one wavefront's registers) ----
    c.wave_blob.clear();
    if (n <= kWaveMaxQubits && n_expr <= kWaveMaxExprs) {
        std::vector<WaveOp> wops;
        std::vector<WaveTerm> wterms;
        std::vector<WaveInit> winit(n, WaveInit{-1, -1});
        std::vector<WaveExpr> wexprs(n_expr);
        for (int e = 0; e < n_expr; ++e) wexprs[e] = WaveExpr{expr_off[e], expr_off[e + 1] - expr_off[e], 1.0};
        // one scale per expression: a program that feeds ONE expression to a half-angle gate (rx / ry / rz) and to a
        // full-angle one (p, the ZZ phase) cannot use this format (the LDS simulator scales per gate, so it runs there)
        std::vector<uint8_t> used_as(n_expr, 0);
        auto half = [&](int e) { if (e >= 0) { wexprs[e].scale = 0.5; used_as[e] |= 1; } };
        auto full = [&](int e) { if (e >= 0) used_as[e] |= 2; };
        std::vector<WaveComp> wcomps;
        std::vector<WaveFactor> wfacts;
        for (const auto& fs : comps) {
            wcomps.push_back(WaveComp{(int32_t)wfacts.size(), (int32_t)fs.size()});
            for (const Factor& f : fs) {
                wfacts.push_back(WaveFactor{f.gate, f.expr});
                if (f.gate != QK_P) half(f.expr); else full(f.expr);
            }
        }
        for (const Item& g : inits) {
            winit[g.q0] = WaveInit{g.gate, g.expr};
            if (g.gate != QK_UC) { if (g.gate != QK_P) half(g.expr); else full(g.expr); }
        }
        for (const Proto& p : seq) {
            if (p.diag) {
                wops.push_back(WaveOp{WV_DIAG, (int32_t)wterms.size(), (int32_t)p.terms.size(), -1});
                for (const Item& t : p.terms) {
                    wterms.push_back(WaveTerm{t.kind, t.q0, t.q1 >= 0 ? t.q1 : 0, t.expr});
                    if (t.kind == TM_RZ) half(t.expr); else full(t.expr);
                }
            } else if (p.nd.kind == ST_U1) {
                wops.push_back(WaveOp{WV_U1, p.nd.q0, p.nd.gate, p.nd.expr});
                if (p.nd.gate != QK_UC) { if (p.nd.gate != QK_P) half(p.nd.expr); else full(p.nd.expr); }
            } else {
                wops.push_back(WaveOp{p.nd.kind == ST_CX ? WV_CX : WV_ECR, p.nd.q0, p.nd.q1, -1});
            }
        }
        auto pad8 = [](size_t b) { return (b + 15) / 16 * 16; };   // 16-byte sections: the blob is staged in LDS by 16-byte copies
        WaveHeader H{};
        H.n_qubits = n;
        H.n_params = c.n_params;
        H.n_ops = (int32_t)wops.size();
        H.n_terms = (int32_t)wterms.size();
        H.n_exprs = n_expr;
        H.n_prog = (int32_t)c.prog.size();
        H.n_consts = (int32_t)c.consts.size();
        H.n_comps = (int32_t)wcomps.size();
        H.spw = n_expr > 0 ? std::max(1, 64 / n_expr) : 64;
        size_t off = sizeof(WaveHeader);
        H.off_ops = (int32_t)off;      off = pad8(off + wops.size() * sizeof(WaveOp));
        H.off_terms = (int32_t)off;    off = pad8(off + wterms.size() * sizeof(WaveTerm));
        H.off_init = (int32_t)off;     off = pad8(off + winit.size() * sizeof(WaveInit));
        H.off_exprs = (int32_t)off;    off = pad8(off + wexprs.size() * sizeof(WaveExpr));
        H.off_prog = (int32_t)off;     off = pad8(off + c.prog.size() * sizeof(qk_instr));
        H.off_consts = (int32_t)off;   off = pad8(off + c.consts.size() * sizeof(double));
        H.off_comps = (int32_t)off;    off = pad8(off + wcomps.size() * sizeof(WaveComp));
        H.off_facts = (int32_t)off;    off = pad8(off + wfacts.size() * sizeof(WaveFactor));
        H.bytes = (int32_t)off;
        bool fits = true;
        for (int e = 0; e < n_expr; ++e) fits = fits && used_as[e] != 3;
        // dynamic LDS of a launch: the (cos, sin) and composite tables of spw samples plus the staged blob (a batch
        // stages the largest blob of its jobs, at most kWaveStageMax); a circuit beyond that runs on the LDS simulator
        const size_t table_bytes = (size_t)16 * std::max(1, H.spw * (H.n_exprs + 4 * H.n_comps));
        fits = fits && table_bytes + (size_t)kWaveStageMax <= (size_t)kWaveLdsLimit;
      if (fits) {
        c.wave_blob.assign(off / 8, 0);
        char* base = (char*)c.wave_blob.data();
        std::memcpy(base, &H, sizeof(H));
        if (!wops.empty()) std::memcpy(base + H.off_ops, wops.data(), wops.size() * sizeof(WaveOp));
        if (!wterms.empty()) std::memcpy(base + H.off_terms, wterms.data(), wterms.size() * sizeof(WaveTerm));
        std::memcpy(base + H.off_init, winit.data(), winit.size() * sizeof(WaveInit));
        if (!wexprs.empty()) std::memcpy(base + H.off_exprs, wexprs.data(), wexprs.size() * sizeof(WaveExpr));
        if (!c.prog.empty()) std::memcpy(base + H.off_prog, c.prog.data(), c.prog.size() * sizeof(qk_instr));
        if (!c.consts.empty()) std::memcpy(base + H.off_consts, c.consts.data(), c.consts.size() * sizeof(double));
        if (!wcomps.empty()) std::memcpy(base + H.off_comps, wcomps.data(), wcomps.size() * sizeof(WaveComp));
        if (!wfacts.empty()) std::memcpy(base + H.off_facts, wfacts.data(), wfacts.size() * sizeof(WaveFactor));
        c.wave_spw = H.spw;
      }
    }

    // ---- 3. passes --------------------------------------------------------
    struct Draft { std::vector<int> S; std::vector<int> stage_ids; int n_term = 0, n_mat = 0; };
    // greedy split for a given number of local qubits; empty result = some gate cannot be placed
    auto make_drafts = [&](int Lcap) {
        const int n_fixed = (n > Lcap) ? std::min(4, n) : 0;  // low qubits kept local: 256-B runs in HBM
        std::vector<Draft> drafts;
        auto new_draft = [&]() {
            Draft d;
            for (int q = 0; q < (n <= Lcap ? n : n_fixed); ++q) d.S.push_back(q);
            drafts.push_back(d);
        };
        auto mats_of = [&](const Item& g) { return g.gate == QK_UC ? 1 + (int)comps[g.expr].size() : 1; };
        new_draft();
        for (const Item& g : inits) drafts.back().n_mat += mats_of(g);
        for (int s = 0; s < (int)seq.size(); ++s) {
            const Proto& p = seq[s];
            std::vector<int> need;
            int add_term = 0, add_mat = 0;
            if (p.diag) {
                add_term = (int)p.terms.size();
            } else {
                if (p.nd.kind == ST_U1) { need = {p.nd.q0}; add_mat = mats_of(p.nd); }
                if (p.nd.kind == ST_CX) need = {p.nd.q1};
                if (p.nd.kind == ST_ECR) need = {p.nd.q0, p.nd.q1};
            }
            for (int attempt = 0; attempt < 2; ++attempt) {
                Draft& d = drafts.back();
                std::vector<int> S2 = d.S;
                for (int q : need)
                    if (std::find(S2.begin(), S2.end(), q) == S2.end()) S2.push_back(q);
                const bool fits = (int)S2.size() <= Lcap && d.n_term + add_term <= kMaxTermPerPass &&
                                  d.n_mat + add_mat <= kMaxMatPerPass;
                if (fits) {
                    d.S = S2;
                    d.stage_ids.push_back(s);
                    d.n_term += add_term;
                    d.n_mat += add_mat;
                    break;
                }
                if (attempt == 1) return std::vector<Draft>();
                new_draft();
            }
        }
        return drafts;
    };
    // 12 local qubits (64 KiB of state) let two workgroups share a CU, so one computes while the
    // other is stuck behind its stores; 13 only where that saves a pass over HBM.
    int Lmax = std::min(n, kMaxLocalQubits);
    std::vector<Draft> drafts = make_drafts(Lmax);
    if (Lmax == kMaxLocalQubits) {
        std::vector<Draft> d12 = make_drafts(kMaxLocalQubits - 1);
        if (!d12.empty() && (drafts.empty() || d12.size() <= drafts.size())) {
            drafts.swap(d12);
            Lmax = kMaxLocalQubits - 1;
        }
    }
    if (drafts.empty()) return fail(QK_ERR_UNSUPPORTED, "cannot place gate in a simulator pass");

    c.passes.clear();
    uint32_t active = 0;  // qubits a non-diagonal gate has acted on so far
    c.stages.clear();
    c.terms.clear();
    c.exprs.clear();
    c.mats.clear();
    c.comps.clear();
    c.facts.clear();
    for (size_t pi = 0; pi < drafts.size(); ++pi) {
        Draft& d = drafts[pi];
        // fill the local set with the lowest unused qubits
        for (int q = 0; q < n && (int)d.S.size() < Lmax; ++q)
            if (std::find(d.S.begin(), d.S.end(), q) == d.S.end()) d.S.push_back(q);
        std::sort(d.S.begin(), d.S.end());
        Pass ps;
        ps.L = Lmax;
        ps.first = (pi == 0);
        // Workgroup size.  The simulator's occupancy is bound by registers (3 waves per SIMD), so what
        // counts is work per wave: one wave per sample up to 9 qubits (the barriers of a one-wave
        // workgroup cost nothing and 12 samples are in flight per CU), two at 10 (measured on gfx950).
        ps.lgT = ps.L >= 13 ? 9 : (ps.L >= 11 ? 8 : (ps.L == 10 ? 7 : 6));
        ps.h = ps.L <= 6 ? ps.L : 6;    // 64 columns = one wavefront; <= kMaxRowBits row bits
        ps.maxg = ps.L >= 12 ? 3 : 2;   // radix-8 sweeps need the 512-thread / 256-VGPR kernel
        // 12 local qubits with butterflies or swaps to do: 8 waves per workgroup on the 128-register kernel (radix-4
        // sweeps).  LDS holds two such states per CU, i.e. four waves per SIMD instead of two -- measured, 4 GiB of Psi:
        // GA genome 3.81 -> 3.15 ms, ZZ r = 2 3.59 -> 3.45 ms, the second pass of 16-qubit ZZ r = 2 5.37 -> 5.1 ms.  A
        // pass that is only diagonal stages (the direct store of the Z / ZZ maps) keeps 4 waves: 0.91 vs 1.19 ms.
        // A pass whose only non-diagonal stages are Hadamards (the layers between the diagonal stages of the Z / ZZ maps
        // at r >= 2) keeps 256 threads on the 256-register kernel and takes them four at a time (radix 16: one item of
        // 16 amplitudes per thread, three sweeps per layer of 12 instead of six of radix 4).
        {
            bool has_nd = false, h_only = true;
            for (int sid : d.stage_ids)
                if (!seq[sid].diag) {
                    has_nd = true;
                    h_only &= seq[sid].nd.kind == ST_U1 && seq[sid].nd.gate == QK_H;
                }
            if (has_nd && h_only && ps.L == 11) { ps.lgT = 7; ps.maxg = 3; }     // 16 amplitudes per thread here too
            if (has_nd && h_only && ps.L == 10) { ps.lgT = 6; ps.maxg = 3; }
            if (has_nd && h_only && ps.L - ps.lgT >= 4) ps.maxh = 4;
            else if (allow_narrow && ps.L == 12 && has_nd) { ps.lgT = 9; ps.maxg = 2; ps.narrow = true; }
        }
        ps.phys = d.S;
        for (int q = 0; q < n; ++q)
            if (std::find(d.S.begin(), d.S.end(), q) == d.S.end()) ps.phys.push_back(q);
        std::vector<int> virt(n);
        for (int v = 0; v < n; ++v) virt[ps.phys[v]] = v;
        // deposit runs (virtual -> physical), maximal runs with constant shift
        ps.nruns = 0;
        for (int v = 0; v < n;) {
            int sh = ps.phys[v] - v, v2 = v;
            uint32_t mask = 0;
            while (v2 < n && ps.phys[v2] - v2 == sh) mask |= (1u << v2), ++v2;
            if (ps.nruns >= kMaxRuns) return fail(QK_ERR_UNSUPPORTED, "local qubit set too fragmented");
            ps.run_mask[ps.nruns] = mask;
            ps.run_shift[ps.nruns] = sh;  // negative for chunk qubits that sit below local ones
            ++ps.nruns;
            v = v2;
        }
        ps.stage_off = (int)c.stages.size();
        ps.expr_off = (int)c.exprs.size();
        ps.mat_off = (int)c.mats.size();
        ps.term_off = (int)c.terms.size();
        ps.comp_off = (int)c.comps.size();
        ps.fact_off = (int)c.facts.size();
        auto add_expr = [&](int e, double scale, int dst, int gate = 0) {
            c.exprs.push_back(DevExpr{expr_off[e], expr_off[e + 1] - expr_off[e], scale, dst, gate});
        };
        auto mask_of = [&](uint32_t qubit_mask) {
            uint32_t m = 0;
            for (int q = 0; q < n; ++q)
                if (qubit_mask >> q & 1u) m |= 1u << virt[q];
            return m;
        };
        // the matrix of a single-qubit item: its own slot, and for a composite one slot per factor before it
        auto add_matrix = [&](const Item& g) {
            if (g.gate != QK_UC) {
                c.mats.push_back(DevMat{g.gate, 0});
                const int mi = (int)c.mats.size() - 1 - ps.mat_off;
                if (g.expr >= 0) add_expr(g.expr, 0.5, ~mi, g.gate);
                return mi;
            }
            DevComp dc{};
            dc.first = (int)c.facts.size() - ps.fact_off;
            dc.count = (int)comps[g.expr].size();
            for (const Factor& f : comps[g.expr]) {
                c.mats.push_back(DevMat{f.gate, 0});
                const int fi = (int)c.mats.size() - 1 - ps.mat_off;
                if (f.expr >= 0) add_expr(f.expr, f.gate == QK_P ? 1.0 : 0.5, ~fi, f.gate);
                c.facts.push_back(fi);
            }
            c.mats.push_back(DevMat{QK_UC, 0});
            dc.dst = (int)c.mats.size() - 1 - ps.mat_off;
            c.comps.push_back(dc);
            return dc.dst;
        };
        if (pi == 0) {
            for (const Item& g : inits) {
                active |= 1u << g.q0;
                const int mi = add_matrix(g);
                DevStage st{};
                st.kind = ST_INIT;
                st.va = virt[g.q0];
                st.gm[0] = mi;
                c.stages.push_back(st);
            }
            ps.n_init = (int)inits.size();
        }
        ps.zero_in = mask_of(~active & ((n >= 32) ? 0xffffffffu : ((1u << n) - 1u)));
        // Hadamards on distinct qubits commute: in a pass that takes them four to a sweep, each layer starts with the
        // four qubits the diagonal sweep walks (virtual bits h .. h + 3), so that the first sweep of the layer can ride
        // on the fused diagonal stage before it (sim_pass_kernel: with_h)
        if (ps.maxh == 4 && ps.L >= ps.h + 4) {
            auto is_h = [&](int sid) { return !seq[sid].diag && seq[sid].nd.kind == ST_U1 && seq[sid].nd.gate == QK_H; };
            for (size_t b = 0; b < d.stage_ids.size();) {
                size_t e = b;
                while (e < d.stage_ids.size() && is_h(d.stage_ids[e])) ++e;
                if (e > b)
                    std::stable_partition(d.stage_ids.begin() + b, d.stage_ids.begin() + e, [&](int sid) {
                        const int v = virt[seq[sid].nd.q0];
                        return v >= ps.h && v < ps.h + 4;
                    });
                b = e > b ? e : b + 1;
            }
        }
        std::vector<int32_t> prev_diag_sig;
        int prev_diag_stage = -1;
        // angle programs by content: every gate carries its own program, the layers of a map repeat the same ones
        auto same_expr = [&](int x, int y) {
            const int lx = expr_off[x + 1] - expr_off[x], ly = expr_off[y + 1] - expr_off[y];
            if (lx != ly) return false;
            for (int k = 0; k < lx; ++k) {
                const qk_instr& p = c.prog[expr_off[x] + k];
                const qk_instr& q = c.prog[expr_off[y] + k];
                if (p.op != q.op) return false;
                if (p.op == QK_EX_CONST ? c.consts[p.arg] != c.consts[q.arg] : (p.op == QK_EX_PARAM && p.arg != q.arg)) return false;
            }
            return true;
        };
        auto canon_expr = [&](int e) {
            if (e < 0) return e;
            for (int y = 0; y < e; ++y)
                if (same_expr(e, y)) return y;
            return e;
        };
        for (int s : d.stage_ids) {
            const Proto& p = seq[s];
            DevStage st{};
            if (p.diag) {
                st.kind = ST_DIAG;
                st.t0 = (int)c.terms.size() - ps.term_off;
                std::vector<std::pair<DevTerm, int>> lo, hi, cross;   // (term, global expr or -1)
                for (const Item& t : p.terms) {
                    DevTerm dt{};
                    dt.type = t.kind;
                    dt.va = virt[t.q0];
                    dt.vb = t.q1 >= 0 ? virt[t.q1] : 0;
                    bool in_lo = false, in_hi = false;
                    auto look = [&](int v) { if (v < ps.h) in_lo = true; else if (v < ps.L) in_hi = true; };
                    look(dt.va);
                    if (t.q1 >= 0) look(dt.vb);
                    (in_lo && in_hi ? cross : (in_hi ? hi : lo)).push_back({dt, t.expr});
                }
                auto row_bit = [&](const DevTerm& t) { return (t.va >= ps.h ? t.va : t.vb) - ps.h; };
                std::stable_sort(cross.begin(), cross.end(),
                                 [&](const auto& x, const auto& y) { return row_bit(x.first) < row_bit(y.first); });
                auto chunk_dep = [&](const std::pair<DevTerm, int>& te) {
                    const DevTerm& t = te.first;
                    const bool two = t.type == TM_CZ || t.type == TM_ZZP;
                    return t.va >= ps.L || (two && t.vb >= ps.L);
                };
                std::stable_partition(lo.begin(), lo.end(), [&](const auto& te) { return !chunk_dep(te); });
                std::stable_partition(hi.begin(), hi.end(), [&](const auto& te) { return !chunk_dep(te); });
                st.nLo = (int)lo.size();
                st.nHi = (int)hi.size();
                st.nX = (int)cross.size();
                st.nLoD = (int)std::count_if(lo.begin(), lo.end(), chunk_dep);
                st.nHiD = (int)std::count_if(hi.begin(), hi.end(), chunk_dep);
                for (int a = 0, k = 0; a <= kMaxRowBits; ++a) {
                    while (k < st.nX && row_bit(cross[k].first) < a) ++k;
                    st.xo[a] = k;
                }
                std::vector<int32_t> sig;      // what the stage's phase tables depend on, in table order
                for (auto* grp : {&lo, &hi, &cross}) {
                    sig.push_back(-1);
                    for (auto& te : *grp) {
                        c.terms.push_back(te.first);
                        if (te.second >= 0)
                            add_expr(te.second, te.first.type == TM_RZ ? 0.5 : 1.0, (int)c.terms.size() - 1 - ps.term_off);
                        sig.insert(sig.end(), {te.first.type, te.first.va, te.first.vb, canon_expr(te.second)});
                    }
                }
                // the layers of a map at r >= 2 repeat the same diagonal stage: its tables are built once
                if (!prev_diag_sig.empty() && sig == prev_diag_sig)
                    st.reuse = (prev_diag_stage == ps.stage_off + ps.n_init && ps.first) ? 2 : 1;
                prev_diag_sig = sig;
                prev_diag_stage = (int)c.stages.size();
                c.stages.push_back(st);
                continue;
            }
            const Item& g = p.nd;
            if (g.kind == ST_U1) {
                active |= 1u << g.q0;
                const int mi = add_matrix(g);
                const int v = virt[g.q0];
                // fuse with the previous stage if that is a U1 group in this pass without this qubit
                bool fused = false;
                if ((int)c.stages.size() > ps.stage_off) {
                    DevStage& prev = c.stages.back();
                    bool all_h = g.gate == QK_H;
                    if (prev.kind == ST_U1)
                        for (int k = 0; k < prev.ng; ++k) all_h &= c.mats[ps.mat_off + prev.gm[k]].gate == QK_H;
                    const int cap = all_h ? std::max(ps.maxg, ps.maxh) : ps.maxg;
                    if (prev.kind == ST_U1 && prev.ng < cap && (int)c.stages.size() > ps.stage_off + ps.n_init) {
                        bool clash = false;
                        for (int k = 0; k < prev.ng; ++k) clash |= (prev.gv[k] == v);
                        if (!clash) {
                            int k = prev.ng++;
                            prev.gv[k] = v;
                            prev.gm[k] = mi;
                            // keep target bits ascending (the gates act on different qubits: they commute)
                            for (; k > 0 && prev.gv[k - 1] > prev.gv[k]; --k) {
                                std::swap(prev.gv[k - 1], prev.gv[k]);
                                std::swap(prev.gm[k - 1], prev.gm[k]);
                            }
                            fused = true;
                        }
                    }
                }
                if (!fused) {
                    st.kind = ST_U1;
                    st.ng = 1;
                    st.gv[0] = v;
                    st.gm[0] = mi;
                    c.stages.push_back(st);
                }
            } else {
                st.kind = g.kind;
                st.va = virt[g.q0];
                st.vb = virt[g.q1];
                active |= 1u << g.q1;                       // cx target / ecr second qubit
                if (g.kind == ST_ECR) active |= 1u << g.q0;
                c.stages.push_back(st);
            }
        }
        if (pi + 1 < drafts.size())
            ps.zero_out = mask_of(~active & ((n >= 32) ? 0xffffffffu : ((1u << n) - 1u)));
        ps.n_stage = (int)c.stages.size() - ps.stage_off;
        // The angle programs of a pass run one per lane, and a wave pays for every distinct operation its lanes are
        // at in a step: programs of one shape (the same operations, other features and constants) side by side
        // keep the interpreter's switch from diverging.  Every entry carries its destination, so the order is free.
        {
            auto shape = [&](const DevExpr& e) {
                uint64_t hsh = 1469598103934665603ull;
                for (int k = 0; k < e.len; ++k) hsh = (hsh ^ (uint64_t)(c.prog[e.start + k].op + 1)) * 1099511628211ull;
                return hsh ^ (uint64_t)e.len;
            };
            std::stable_sort(c.exprs.begin() + ps.expr_off, c.exprs.end(),
                             [&](const DevExpr& x, const DevExpr& y) {
                                 if (x.len != y.len) return x.len > y.len;     // long programs first: they share the early steps
                                 return shape(x) < shape(y);
                             });
        }
        ps.n_expr = (int)c.exprs.size() - ps.expr_off;
        ps.n_term = (int)c.terms.size() - ps.term_off;
        ps.n_mat = (int)c.mats.size() - ps.mat_off;
        ps.n_comp = (int)c.comps.size() - ps.comp_off;
        c.passes.push_back(ps);
    }

    // ---- 4. the register-resident program (one pass, every qubit local: qk_reg.h) ------------------------------
    // 10 to 13 qubits: one sample per workgroup.  4 to 9 qubits: 2^(Lv - n) samples per workgroup side by side in a
    // "virtual" register of Lv = 10 .. 12 index bits whose top bits number the sample -- they are thread bits in every
    // layout and no gate touches them, so one instruction stream (segments, ops, transposes) serves all those samples
    // and only the angles, matrices and phase tables exist once per sample.
    c.reg_blob.clear();
    bool q_tables = true;      // register-thread terms of diagonal stages as ratio tables (below)
    static const bool merge_tables = !(std::getenv("QK_REG_MERGE") && std::atoi(std::getenv("QK_REG_MERGE")) == 0);      // (A/B runs)
    auto plan_reg = [&](const int Lv, const size_t lds_limit, size_t* dry_lds = nullptr) -> bool {      // dry_lds: only report the LDS a workgroup would need
        // Single-qubit gates on distinct qubits with nothing between them commute: inside such a run the gates are
        // taken in ascending qubit order, so that a layer of gates fills whole groups of four neighbouring qubits and
        // the groups with the low qubits come first (the last layout then keeps the low qubits in the thread index:
        // the final stores cover consecutive amplitudes).
        std::vector<int> order(seq.size());
        for (size_t k = 0; k < seq.size(); ++k) order[k] = (int)k;
        auto is_u1 = [&](int id) { return !seq[id].diag && seq[id].nd.kind == ST_U1; };
        for (size_t b = 0; b < order.size();) {
            size_t e = b;
            uint32_t used = 0;
            while (e < order.size() && is_u1(order[e]) && !((used >> seq[order[e]].nd.q0) & 1u)) used |= 1u << seq[order[e++]].nd.q0;
            if (e > b + 1) std::stable_sort(order.begin() + b, order.begin() + e, [&](int x, int y) { return seq[x].nd.q0 < seq[y].nd.q0; });
            b = e > b ? e : b + 1;
        }
        // Segments: consecutive gates whose non-diagonal qubits fit the RB register qubits of one layout.
        struct SegDraft { std::vector<int> bits, ids; };
        auto draft = [&](const int RB) {
            std::vector<SegDraft> sd(1);
            for (int id : order) {
                const Proto& p = seq[id];
                std::vector<int> need;
                if (!p.diag) {
                    if (p.nd.kind == ST_U1) need = {p.nd.q0};
                    else if (p.nd.kind == ST_CX) need = {p.nd.q1};
                    else need = {p.nd.q0, p.nd.q1};
                }
                std::vector<int> b2 = sd.back().bits;
                for (int q : need)
                    if (std::find(b2.begin(), b2.end(), q) == b2.end()) b2.push_back(q);
                if ((int)b2.size() > RB) {
                    sd.push_back(SegDraft{});
                    b2 = need;
                }
                sd.back().bits = b2;
                sd.back().ids.push_back(id);
            }
            return sd;
        };
        // Register qubits per layout: 4 = 16 amplitudes per thread, 3 = 8 amplitudes and twice the threads.  With 8 a
        // layer of gates needs more layouts (a transpose through LDS each) and a diagonal stage computes its per-thread
        // phase for half as many amplitudes; with 16 a CU holds half the waves, and the butterflies with a matrix, the
        // swaps and the ECRs are chains the other waves have to cover.  Measured (4 GiB of Psi per call, 16 -> 8):
        //   7 qubits and below, every circuit: 8 wins (6 qubits Z r = 2 3.93 -> 2.62 ms, 5 qubits custom1 r = 2 32.4 -> 19.7);
        //   GA genomes: 8 wins (12 qubits 1.89 -> 1.76, 11: 2.55 -> 2.13, 10: 2.07 -> 1.52, 9: 3.18 -> 2.63, 8: 2.82 -> 2.31);
        //   the Z / ZZ maps (diagonal stages and Hadamard layers only): 16 wins (12 qubits ZZ r = 2 1.62 against 1.74);
        //   custom1 r = 2 (heavy diagonal stages and one layer of rotations): 16 wins where 8 costs a layout more
        //   (8 qubits 4.19 against 4.97, 10: 2.60 / 2.73, 12: 2.03 / 2.15), 8 where it does not (9 qubits 5.33 -> 3.96).
        // "busy": gates that are butterflies with a matrix, swaps or ECRs -- not the first gate on a qubit, which goes into
        // the product state, and not Hadamards, whose layers are sweeps of additions.  QK_REG_RB overrides (A/B runs).
        static const int rb_env = std::getenv("QK_REG_RB") ? std::atoi(std::getenv("QK_REG_RB")) : 0;
        int busy = 0, n_diag_terms = 0;
        {
            uint32_t seen = 0;
            for (const Proto& p : seq) {
                if (p.diag) {
                    n_diag_terms += (int)p.terms.size();
                    for (const Item& t : p.terms) seen |= (1u << t.q0) | (t.q1 >= 0 ? 1u << t.q1 : 0u);
                    continue;
                }
                const bool u1 = p.nd.kind == ST_U1;
                if (u1 && !((seen >> p.nd.q0) & 1u)) { seen |= 1u << p.nd.q0; continue; }
                if (!(u1 && p.nd.gate == QK_H)) ++busy;
                seen |= (1u << p.nd.q0) | (p.nd.q1 >= 0 ? 1u << p.nd.q1 : 0u);
            }
        }
        int RB = 4;
        if (!allow_narrow && Lv == n) RB = 4;      // the plan of the reduced-density-matrix mode (qk_circuit::rdm_plan): its reduction needs the 16-amplitude form
        else if (rb_env == 3 || rb_env == 4) RB = rb_env;
        else if (n <= 7) RB = 3;
        else if (2 * busy >= n) RB = (n_diag_terms >= 2 * n && draft(3).size() > draft(4).size()) ? 4 : 3;
        const int n_samp = 1 << (Lv - n);      // samples per workgroup
        std::vector<SegDraft> sd = draft(RB);
        for (SegDraft& g : sd) {      // fill the group with the highest qubits it does not hold yet
            for (int q = n - 1; q >= 0 && (int)g.bits.size() < RB; --q)
                if (std::find(g.bits.begin(), g.bits.end(), q) == g.bits.end()) g.bits.push_back(q);
            std::sort(g.bits.begin(), g.bits.end());
        }
        std::vector<RegSeg> segs;
        std::vector<RegOp> rops;
        std::vector<int> diag_hs;          // per diagonal op, in order: slots of its high-bits table
        std::vector<RegTab> tabs;
        std::vector<RegTerm> rterms;      // terms the simulator evaluates per thread (between two thread bits, or a register and a thread bit)
        struct TabTerm { int type, va, vb, tag; };
        std::vector<TabTerm> tterms;      // terms that live in a phase table (reg_table_kernel)
        // DevExpr::dst of an angle that only the pre-pass kernels read (table terms, factors of matrices; DevExpr::gate = its
        // entry in pre_users); every other dst is a term of `rterms`.  Equal angles (the same program at the same scale:
        // every repetition of a map re-uses the angles of the first) are evaluated once.
        constexpr int kPreDst = 0x7fffffff;
        struct PreUser { bool table; int index; };
        std::vector<std::vector<PreUser>> pre_users;
        int n_itab = 0;
        std::vector<DevExpr> rexprs;
        std::vector<RegMat> rmats;        // the matrices the simulator reads, as products of factors (reg_matrix_kernel)
        std::vector<RegFact> rfacts;
        std::vector<int32_t> init_mat(n, -1);
        auto r_expr = [&](int e, double scale, int dst, int gate = 0) {
            rexprs.push_back(DevExpr{expr_off[e], expr_off[e + 1] - expr_off[e], scale, dst, gate});
        };
        auto r_pre = [&](int e, double scale, PreUser user) {
            const int start = expr_off[e], len = expr_off[e + 1] - expr_off[e];
            for (const DevExpr& x : rexprs) {
                if (x.dst != kPreDst || x.len != len || x.scale != scale) continue;
                bool same = true;
                for (int k = 0; k < len && same; ++k) same = c.prog[x.start + k].op == c.prog[start + k].op && c.prog[x.start + k].arg == c.prog[start + k].arg;
                if (same) { pre_users[x.gate].push_back(user); return; }
            }
            pre_users.push_back({user});
            r_expr(e, scale, kPreDst, (int)pre_users.size() - 1);
        };
        auto r_factor = [&](int gate, int expr) {
            rfacts.push_back(RegFact{gate, -1});
            if (expr >= 0) r_pre(expr, gate == QK_P ? 1.0 : 0.5, PreUser{false, (int)rfacts.size() - 1});
        };
        auto r_matrix = [&](const Item& g) {
            RegMat m{(int)rfacts.size(), 1};
            if (g.gate != QK_UC) {
                r_factor(g.gate, g.expr);
            } else {
                m.count = (int)comps[g.expr].size();
                for (const Factor& f : comps[g.expr]) r_factor(f.gate, f.expr);
            }
            rmats.push_back(m);
            return (int)rmats.size() - 1;
        };
        bool all_h = (int)inits.size() == n;
        for (const Item& g : inits) all_h = all_h && g.gate == QK_H;
        if (!all_h)      // (an all-Hadamard first layer is the constant 2^(-n/2): no matrices)
            for (const Item& g : inits) init_mat[g.q0] = r_matrix(g);
        auto table = [&](const std::vector<int>& bits, size_t from, uint32_t fixed, int t0, int nt) {
            RegTab t{};
            t.nbits = (int)std::min<size_t>(4, bits.size() - std::min(from, bits.size()));
            for (int k = 0; k < t.nbits; ++k) t.bits[k] = bits[from + k];
            t.fixed = fixed;
            t.t0 = t0;
            t.nt = nt;
            tabs.push_back(t);
        };
        for (size_t si = 0; si < sd.size(); ++si) {
            const SegDraft& g = sd[si];
            RegSeg sg{};
            std::vector<int> tb, lo, hi;
            for (int k = 0; k < RB; ++k) sg.g[k] = g.bits[k];
            auto rpos = [&](int q) { return (int)(std::find(g.bits.begin(), g.bits.end(), q) - g.bits.begin()); };
            for (int q = 0; q < n; ++q)
                if (rpos(q) == RB) tb.push_back(q);
            {   // deposit runs: thread bit j -> qubit tb[j], the sample bits above them -> index bits n .. Lv - 1;
                // maximal runs with a constant shift
                std::vector<int> dep = tb;
                for (int q = n; q < Lv; ++q) dep.push_back(q);
                sg.nrun = 0;
                for (size_t j = 0; j < dep.size();) {
                    const int sh = dep[j] - (int)j;
                    uint32_t mask = 0;
                    while (j < dep.size() && dep[j] - (int)j == sh) mask |= 1u << j, ++j;
                    sg.run[sg.nrun++] = mask | ((uint32_t)sh << 16);
                }
            }
            lo.assign(tb.begin(), tb.begin() + std::min<size_t>(4, tb.size()));
            hi.assign(tb.begin() + lo.size(), tb.end());
            auto in = [](const std::vector<int>& v, int q) { return std::find(v.begin(), v.end(), q) != v.end(); };
            // three tables (four when the thread has nine bits): register bits, low thread bits, high thread bits
            // three tables: register bits, low thread bits, high thread bits -- the last one in 2^(extra) slots of 16
            // entries when the thread has more than eight bits (slot = value of the extra bits, ascending)
            auto thread_tables = [&](int tL, int nL, int tH, int nH) {
                table(lo, 0, 0u, tL, nL);
                const int extra = std::max(0, (int)hi.size() - 4);
                for (int x = 0; x < (1 << extra) && !hi.empty(); ++x) {     // (no high thread qubits: no such table)
                    uint32_t fixed = 0;
                    for (int b = 0; b < extra; ++b)
                        if ((x >> b) & 1) fixed |= 1u << hi[4 + b];
                    table(hi, 0, fixed, tH, nH);
                }
            };
            auto tables = [&](int tA, int nA, int tL, int nL, int tH, int nH) {
                const int first = (int)tabs.size();
                table(g.bits, 0, 0u, tA, nA);
                thread_tables(tL, nL, tH, nH);
                return first;
            };
            uint32_t extra_mask = 0;
            for (size_t b = 4; b < hi.size(); ++b) extra_mask |= 1u << hi[b];
            sg.op0 = (int)rops.size();
            if (si == 0) {
                RegOp op{};
                op.kind = RG_INIT;
                op.flags = all_h ? 1 : 0;
                op.tab = tables(0, -1, 0, -1, (int)extra_mask, -1);   // nt < 0: first-column products; t0 = mask of the extra qubits (values: fixed)
                n_itab = (int)tabs.size();
                rops.push_back(op);
            }
            for (int id : g.ids) {
                const Proto& p = seq[id];
                RegOp op{};
                if (p.diag) {
                    std::vector<Item> cls[5];     // RR, lo, hi, TX, RT
                    for (const Item& t : p.terms) {
                        const bool two = t.q1 >= 0 && (t.kind == TM_CZ || t.kind == TM_ZZP);
                        const bool r0 = in(g.bits, t.q0), r1 = two && in(g.bits, t.q1);
                        int k;
                        if (!two) k = r0 ? 0 : (in(lo, t.q0) ? 1 : 2);
                        else if (r0 && r1) k = 0;
                        else if (r0 || r1) k = 4;
                        else if (in(lo, t.q0) && in(lo, t.q1)) k = 1;
                        else if (in(hi, t.q0) && in(hi, t.q1)) k = 2;
                        else k = 3;
                        cls[k].push_back(t);
                    }
                    // A term between a register qubit and a thread qubit (class RT), as tables: what the term is while the
                    // register bit is 0 depends on the thread bit alone and joins that bit's thread table; what flipping
                    // the register bit multiplies the phase by -- exp(+-i theta) by the thread bit for a ZZ term, -1 where
                    // the thread bit is set for a CZ -- goes into a pair of "ratio" tables per register position (low and
                    // high thread bits), which the kernel multiplies into its Gray-code walk.  Evaluated per thread instead
                    // (q_tables off), every such term costs each thread two factors and three complex products.
                    constexpr int kRatio = 0x100;
                    std::vector<Item> ratio[4][2];      // [register position][low / high thread bits]
                    // Worth it from three terms per register position (a map with full entanglement: 8 per position at 12
                    // qubits, ZZ r = 2 4.78 -> 2.27 ms per 4 GiB); the linear maps have one or two such terms per stage,
                    // which cost less than the tables' place in LDS (10 qubits ZZ r = 2: 1.94 ms without, 2.12 with).
                    uint32_t rt_pos = 0;
                    for (const Item& t : cls[4]) rt_pos |= 1u << rpos(in(g.bits, t.q0) ? t.q0 : t.q1);
                    if (q_tables && (int)cls[4].size() >= 3 * __builtin_popcount(rt_pos) && !cls[4].empty()) {
                        for (const Item& t : cls[4]) {
                            const bool reg0 = in(g.bits, t.q0);
                            const int rq = reg0 ? t.q0 : t.q1, tq = reg0 ? t.q1 : t.q0;
                            const int half = in(lo, tq) ? 0 : 1;
                            Item r = t;
                            r.q0 = tq; r.q1 = -1;
                            ratio[rpos(rq)][half].push_back(r);
                            op.flags |= (1 << rpos(rq)) | 512;
                            if (t.kind == TM_ZZP) {      // register bit 0: exp(i theta) where the thread bit is set -- a phase gate on it
                                Item f0 = t;
                                f0.kind = TM_P; f0.q0 = tq; f0.q1 = -1;
                                cls[1 + half].push_back(f0);
                            }
                        }
                        cls[4].clear();
                    }
                    auto push_tterm = [&](const Item& t, int type) {
                        tterms.push_back(TabTerm{type, t.q0, t.q1 >= 0 ? t.q1 : 0, -1});
                        if (t.expr >= 0) {
                            tterms.back().tag = (int)tterms.size() - 1;
                            r_pre(t.expr, t.kind == TM_RZ ? 0.5 : 1.0, PreUser{true, (int)tterms.size() - 1});
                        }
                    };
                    int tstart[4];      // RR, lo, hi -> table terms
                    for (int k = 0; k < 3; ++k) {
                        tstart[k] = (int)tterms.size();
                        for (const Item& t : cls[k]) push_tterm(t, t.kind);
                    }
                    tstart[3] = (int)tterms.size();
                    int qstart[4][3];
                    for (int rp = 0; rp < 4; ++rp) {
                        for (int h = 0; h < 2; ++h) {
                            qstart[rp][h] = (int)tterms.size();
                            for (const Item& t : ratio[rp][h]) push_tterm(t, t.kind | kRatio);
                        }
                        qstart[rp][2] = (int)tterms.size();
                    }
                    int start[6] = {0, 0, 0, 0, 0, 0};      // TX, RT -> evaluated per thread
                    for (int k = 3; k < 5; ++k) {
                        start[k] = (int)rterms.size();
                        for (const Item& t : cls[k]) {
                            RegTerm rt{};
                            rt.type = t.kind;
                            rt.va = t.q0;
                            rt.vb = t.q1 >= 0 ? t.q1 : 0;
                            rt.expr = -1;
                            if (k == 4) {
                                const int rp = in(g.bits, t.q0) ? rpos(t.q0) : rpos(t.q1);
                                rt.type |= rp << 8;
                                op.flags |= 1 << rp;
                            }
                            rterms.push_back(rt);
                            if (t.expr >= 0) r_expr(t.expr, t.kind == TM_RZ ? 0.5 : 1.0, (int)rterms.size() - 1);
                        }
                    }
                    start[5] = (int)rterms.size();
                    op.kind = RG_DIAG;
                    // directly behind the product state of an all-Hadamard first layer every amplitude is still the constant
                    // held in the running scale: the stage writes its phases instead of multiplying by them
                    if (!rops.empty() && (int)rops.size() == sg.op0 + 1 && rops.back().kind == RG_INIT && (rops.back().flags & 1)) op.flags |= 256;
                    op.tab = tables(tstart[0], tstart[1] - tstart[0], tstart[1], tstart[2] - tstart[1], tstart[2], tstart[3] - tstart[2]) - n_itab;
                    const int hs = hi.empty() ? 0 : 1 << std::max(0, (int)hi.size() - 4);      // slots of the high-bits table
                    op.a = op.tab + 1;                 // Blo, Bhi: consecutive for now; equal tables are merged below
                    op.b = op.tab + 2;
                    op.tab |= (op.tab + 2 + hs) << 16; // the ratio tables of the register positions that have any, in order of position
                    if (op.flags & 512)
                        for (int rp = 0; rp < RB; ++rp)
                            if ((op.flags >> rp) & 1) thread_tables(qstart[rp][0], qstart[rp][1] - qstart[rp][0], qstart[rp][1], qstart[rp][2] - qstart[rp][1]);
                    diag_hs.push_back(hs);
                    op.t0 = start[3];
                    op.nTX = start[4] - start[3];
                    op.nRT = start[5] - start[4];
                } else if (p.nd.kind == ST_U1) {
                    // gates on distinct register qubits commute: up to four of them travel as ONE op (one record, the
                    // matrices read together) -- a layer of Hadamards on the group is a single op without matrices
                    const int pos = rpos(p.nd.q0), mi = p.nd.gate == QK_H ? 0 : r_matrix(p.nd);      // (a Hadamard is adds and subtractions)
                    const int gate = p.nd.gate;      // butterfly form: 3 hadamard (adds only), 0 real matrix, 1 rx-shaped, 2 general
                    const int form = gate == QK_H ? 3 : ((gate == QK_X || gate == QK_RY) ? 0 : (gate == QK_RX ? 1 : 2));
                    if (!rops.empty() && (int)rops.size() > sg.op0 && rops.back().kind == RG_U1 && !((rops.back().a >> pos) & 1)) {
                        RegOp& u = rops.back();
                        u.a |= 1 << pos;
                        u.b |= mi << (8 * pos);
                        u.flags |= form << (2 * pos);
                        continue;
                    }
                    op.kind = RG_U1;
                    op.a = 1 << pos;
                    op.b = mi << (8 * pos);
                    op.flags = form << (2 * pos);
                } else if (p.nd.kind == ST_CX) {
                    op.b = rpos(p.nd.q1);
                    if (in(g.bits, p.nd.q0)) { op.kind = RG_CX; op.a = rpos(p.nd.q0); }
                    else { op.kind = RG_CXT; op.a = (int)(std::find(tb.begin(), tb.end(), p.nd.q0) - tb.begin()); }
                } else {
                    op.kind = RG_ECR;
                    op.a = rpos(p.nd.q0);
                    op.b = rpos(p.nd.q1);
                }
                rops.push_back(op);
            }
            sg.op1 = (int)rops.size();
            segs.push_back(sg);
        }
        // angle programs the simulator itself reads first (terms per thread, matrices), those of the table kernel behind
        // them; inside each part programs of one shape side by side (as for the LDS kernel's pre-pass: uniform control flow)
        auto shape = [&](const DevExpr& e) {
            uint64_t hsh = 1469598103934665603ull;
            for (int k = 0; k < e.len; ++k) hsh = (hsh ^ (uint64_t)(c.prog[e.start + k].op + 1)) * 1099511628211ull;
            return hsh ^ (uint64_t)e.len;
        };
        auto pre_only = [&](const DevExpr& e) { return e.dst == kPreDst; };
        std::stable_sort(rexprs.begin(), rexprs.end(), [&](const DevExpr& x, const DevExpr& y) {
            if (pre_only(x) != pre_only(y)) return pre_only(y);
            if (x.len != y.len) return x.len > y.len;
            return shape(x) < shape(y);
        });
        size_t n_expr_k = 0;
        std::vector<int32_t> term_expr(rterms.size(), -1);
        for (size_t e = 0; e < rexprs.size(); ++e) {      // where the terms and factors find their (cos, sin)
            if (rexprs[e].dst == kPreDst) {
                for (const PreUser& u : pre_users[rexprs[e].gate]) (u.table ? tterms[u.index].tag : rfacts[u.index].expr) = (int)e;
            } else {
                term_expr[rexprs[e].dst] = (int)e;
                n_expr_k = e + 1;
            }
        }
        for (size_t t = 0; t < rterms.size(); ++t) rterms[t].expr = term_expr[t];
        // the table kernel's records: which entries of a table take a term's factor, which its conjugate
        std::vector<RegTabTerm> tts;
        for (size_t ti = (size_t)n_itab; ti < tabs.size(); ++ti) {
            RegTab& tb = tabs[ti];
            const int first = (int)tts.size();
            for (int k = 0; k < tb.nt; ++k) {
                const TabTerm& t = tterms[tb.t0 + k];
                RegTabTerm r{(t.type & 0xff) == TM_CZ ? -1 : t.tag, 0u};
                for (uint32_t x = 0; x < (1u << tb.nbits); ++x) {
                    uint32_t w = tb.fixed;
                    for (int b = 0; b < tb.nbits; ++b) w |= ((x >> b) & 1u) << tb.bits[b];
                    const uint32_t ba = (w >> t.va) & 1u, bb = (w >> t.vb) & 1u;
                    if (t.type & 0x100) {      // a ratio table's term (va: the thread qubit)
                        if ((t.type & 0xff) == TM_CZ) { if (ba) r.masks |= 1u << x; }
                        else r.masks |= (1u << x) | (ba ? 1u << (16 + x) : 0u);      // exp(i theta) / exp(-i theta)
                        continue;
                    }
                    const bool on = t.type == TM_RZ || (t.type == TM_P ? ba != 0u : (t.type == TM_CZ ? (ba & bb) != 0u : (ba ^ bb) != 0u));
                    if (on) r.masks |= 1u << x;
                    if (t.type == TM_RZ && !ba) r.masks |= 1u << (16 + x);
                }
                tts.push_back(r);
            }
            tb.t0 = first;
        }
        // Equal tables exist once: the repetitions of a map apply the same diagonal stage in different layouts, and what was
        // the register table of one is a thread table of the other (ZZ r = 2 at 8 qubits: four stage tables per sample
        // become two -- a third less for the table kernel to write, for the simulator to fetch and to hold in LDS).
        // A table = its bits' count and its (angle, masks) records; high-bits tables compare slot group by slot group.
        if (merge_tables) {
            auto same = [&](int x, int y, int count) {
                for (int k = 0; k < count; ++k) {
                    const RegTab &tx = tabs[n_itab + x + k], &ty = tabs[n_itab + y + k];
                    if (tx.nbits != ty.nbits || tx.nt != ty.nt) return false;
                    for (int t = 0; t < tx.nt; ++t)
                        if (tts[tx.t0 + t].expr != tts[ty.t0 + t].expr || tts[tx.t0 + t].masks != tts[ty.t0 + t].masks) return false;
                }
                return true;
            };
            const int n_d = (int)tabs.size() - n_itab;
            std::vector<int> keep(n_d, 1), target(n_d, -1);      // target: the earlier table a dropped one is replaced by
            struct Group { int first, count; };
            std::vector<Group> groups;      // the A / Blo / Bhi groups seen so far (ratio tables are left alone)
            int di = 0;
            for (RegOp& o : rops) {
                if (o.kind != RG_DIAG) continue;
                const int hs = diag_hs[di++];
                int* idx[3] = {nullptr, &o.a, &o.b};
                const int first[3] = {o.tab & 0xffff, o.a, o.b}, count[3] = {1, 1, hs};
                for (int k = 0; k < 3; ++k) {
                    if (count[k] == 0) continue;
                    int hit = -1;
                    for (const Group& g : groups)
                        if (g.count == count[k] && same(g.first, first[k], count[k])) { hit = g.first; break; }
                    if (hit < 0) { groups.push_back(Group{first[k], count[k]}); continue; }
                    for (int j = 0; j < count[k]; ++j) { keep[first[k] + j] = 0; target[first[k] + j] = hit + j; }
                    if (k == 0) o.tab = (o.tab & ~0xffff) | hit; else *idx[k] = hit;
                }
            }
            std::vector<int> remap(n_d + 1, 0);
            int n_new = 0;
            for (int t = 0; t < n_d; ++t) { remap[t] = n_new; n_new += keep[t]; }
            remap[n_d] = n_new;
            auto moved = [&](int t) { return t < n_d && !keep[t] ? remap[target[t]] : remap[t]; };
            for (RegOp& o : rops) {
                if (o.kind != RG_DIAG) continue;
                o.tab = moved(o.tab & 0xffff) | (remap[(o.tab >> 16) & 0xffff] << 16);
                o.a = moved(o.a);
                o.b = moved(o.b);
            }
            std::vector<RegTab> kept(tabs.begin(), tabs.begin() + n_itab);
            for (int t = 0; t < n_d; ++t)
                if (keep[t]) kept.push_back(tabs[n_itab + t]);
            tabs.swap(kept);
        }
        const bool fits = (int)tabs.size() <= kRegMaxTabs && (int)rterms.size() <= kRegMaxTerms && (int)rmats.size() <= std::min(kRegMaxMats, 255);
        // LDS: the transpose image; per sample the product-state tables (unless all-Hadamard);
        // two buffers each of a group's stage tables + matrices (from the pre-pass kernels) and of its angles; staged records
        const size_t n_dtab = tabs.size() - (size_t)n_itab, n_itab_built = all_h ? 0 : (size_t)n_itab;
        const size_t n_slot = ((size_t)n_samp * n_expr_k + 63) / 64 * 64;      // slots of 64 (cos, sin) pairs
        const size_t side_piece = (size_t)n_samp * (n_dtab * 16 + rmats.size() * 4);      // 16-byte pieces of a group's tables and matrices
        const size_t t_slot = (side_piece + 63) / 64;
        // (the tables and matrices of the NEXT group arrive while this one computes: two buffers -- one, filled at the top
        // of a group's turn, when two do not fit even a lone workgroup per CU: deep circuits at 13 qubits)
        const size_t lds_but_side = ((size_t)16 << Lv) + (size_t)n_samp * 256 * n_itab_built + 32 * (size_t)n_itab +
                                    4 * (size_t)n + 64 + 32 * n_slot;
        const bool tab_single = lds_but_side + 2048 * t_slot > lds_limit && lds_limit >= (size_t)160 * 1024;
        const size_t lds = lds_but_side + (tab_single ? 1024 : 2048) * t_slot;
        const size_t waves = std::max<size_t>(1, ((size_t)1 << (Lv - RB)) / 64);
        const size_t nd = (n_slot / 64 + waves - 1) / waves + (t_slot + waves - 1) / waves;    // fetch instructions per wave and sample group
        if (std::getenv("QK_REG_DEBUG")) std::fprintf(stderr, "[reg plan] Lv=%d fits=%d nd=%zu lds=%zu limit=%zu tabs=%zu terms=%zu mats=%zu\n", Lv, (int)fits, nd, lds, lds_limit, tabs.size(), rterms.size(), rmats.size());
        if (!(fits && nd <= 64 && lds <= lds_limit)) return false;
        if (dry_lds) {      // (bit 40: product-state tables are built in the kernel; bit 41: 8 amplitudes per thread; bit 42: 2n or more per-thread terms)
            *dry_lds = lds | (all_h ? (size_t)0 : (size_t)1 << 40) | (RB == 3 ? (size_t)1 << 41 : (size_t)0) |
                       (rterms.size() >= (size_t)(2 * n) ? (size_t)1 << 42 : (size_t)0);
            return true;
        }
        {
            auto pad16 = [](size_t b) { return (b + 15) / 16 * 16; };
            RegHeader H{};
            H.L = Lv;
            H.nq = n;
            if (const char* dbg = std::getenv("QK_REG_DEBUG"); dbg && std::atoi(dbg) >= 2) {      // the layouts and their ops, one line per segment
                static const char* const kind_name[] = {"init", "diag", "u1", "cx", "cxt", "ecr"};
                for (const RegSeg& sg : segs) {
                    std::fprintf(stderr, "[reg plan]   layout {%d %d %d %d}:", sg.g[0], sg.g[1], sg.g[2], sg.g[3]);
                    for (int k = sg.op0; k < sg.op1; ++k) {
                        const RegOp& o = rops[(size_t)k];
                        if (o.kind == RG_DIAG) std::fprintf(stderr, " diag(tx %d rt %d%s)", o.nTX, o.nRT, (o.flags & 512) ? " ratio" : "");
                        else if (o.kind == RG_U1) std::fprintf(stderr, " u1(mask %x)", (unsigned)o.a);
                        else std::fprintf(stderr, " %s(%d,%d)", kind_name[o.kind], o.a, o.b);
                    }
                    std::fprintf(stderr, "\n");
                }
            }
            H.n_seg = (int)segs.size(); H.n_op = (int)rops.size(); H.n_tab = (int)tabs.size(); H.n_term = (int)rterms.size();
            H.n_expr = (int)rexprs.size(); H.n_mat = (int)rmats.size(); H.n_comp = 0; H.n_fact = (int)rfacts.size();
            H.init_h = (all_h ? 1 : 0) | (tab_single ? 2 : 0);
            H.rb = RB;
            H.init_c = std::ldexp((n & 1) ? 0.70710678118654752440 : 1.0, -(n / 2));
            H.n_itab = n_itab; H.n_tt = (int)tts.size(); H.n_expr_k = (int)n_expr_k;
            size_t off = pad16(sizeof(RegHeader));
            H.off_seg = (int)off;  off = pad16(off + segs.size() * sizeof(RegSeg));
            H.off_op = (int)off;   off = pad16(off + rops.size() * sizeof(RegOp));
            H.off_tab = (int)off;  off = pad16(off + tabs.size() * sizeof(RegTab));
            H.off_term = (int)off; off = pad16(off + rterms.size() * sizeof(RegTerm));
            H.off_expr = (int)off; off = pad16(off + rexprs.size() * sizeof(DevExpr));
            H.off_mat = (int)off;  off = pad16(off + rmats.size() * sizeof(RegMat));
            H.off_comp = (int)off;
            H.off_fact = (int)off; off = pad16(off + rfacts.size() * sizeof(RegFact));
            H.off_init = (int)off; off = pad16(off + init_mat.size() * sizeof(int32_t));
            H.off_tt = (int)off;   off = pad16(off + tts.size() * sizeof(RegTabTerm));
            H.bytes = (int)off;
            c.reg_blob.assign(off / 8, 0);
            char* base = (char*)c.reg_blob.data();
            auto put = [&](int at, const void* src, size_t bytes) { if (bytes) std::memcpy(base + at, src, bytes); };
            put(0, &H, sizeof(H));
            put(H.off_seg, segs.data(), segs.size() * sizeof(RegSeg));
            put(H.off_op, rops.data(), rops.size() * sizeof(RegOp));
            put(H.off_tab, tabs.data(), tabs.size() * sizeof(RegTab));
            put(H.off_term, rterms.data(), rterms.size() * sizeof(RegTerm));
            put(H.off_expr, rexprs.data(), rexprs.size() * sizeof(DevExpr));
            put(H.off_mat, rmats.data(), rmats.size() * sizeof(RegMat));
            put(H.off_fact, rfacts.data(), rfacts.size() * sizeof(RegFact));
            put(H.off_init, init_mat.data(), init_mat.size() * sizeof(int32_t));
            put(H.off_tt, tts.data(), tts.size() * sizeof(RegTabTerm));
            c.reg_lds_bytes = lds;
            c.plan_summary += "; register-resident plan: " + std::to_string(segs.size()) + " segments, " + std::to_string(rops.size()) +
                              " ops, " + std::to_string(tabs.size()) + " tables, " + std::to_string(n_samp) + " samples per workgroup of " +
                              std::to_string(1 << (Lv - RB)) + " threads";
            int n_ratio = 0;
            for (const RegOp& o : rops)
                if (o.kind == RG_DIAG && (o.flags & 512)) n_ratio += __builtin_popcount((unsigned)o.flags & 15u);
            if (n_ratio) c.plan_summary += ", ratio tables for " + std::to_string(n_ratio) + " register positions";
        }
        return true;
    };
    // ... with the ratio tables of the register-thread terms unless their LDS costs more than a third of the resident
    // workgroups (QK_REG_QTAB=0 / 1: A/B runs)
    auto plan_q = [&](const int Lv, const size_t lds_limit, size_t* dry_lds = nullptr) -> bool {
        static const int q_env = std::getenv("QK_REG_QTAB") ? std::atoi(std::getenv("QK_REG_QTAB")) : -1;
        if (!two_qubit_diag) {      // (no term between two qubits: nothing a ratio table could take, one draft is enough)
            q_tables = false;
            return plan_reg(Lv, lds_limit, dry_lds);
        }
        size_t l1 = 0, l0 = 0;
        q_tables = true;
        const bool ok1 = q_env != 0 && plan_reg(Lv, lds_limit, &l1);
        q_tables = false;
        const bool ok0 = q_env != 1 && plan_reg(Lv, lds_limit, &l0);
        if (!ok1 && !ok0) return false;
        auto per_cu = [](size_t l) { return ((size_t)160 * 1024) / (l & (((size_t)1 << 40) - 1)); };
        q_tables = ok1 && (!ok0 || 3 * per_cu(l1) >= 2 * per_cu(l0));      // (10 qubits ZZ r = 2, full entanglement: 6 workgroups per CU instead of 7, 5.11 -> 3.22 ms)
        return plan_reg(Lv, lds_limit, dry_lds);
    };
    // (one pass of the LDS plan: at 13 qubits that plan prefers 12 local qubits and a chunk bit where a qubit has no butterfly
    // -- most GA genomes -- which is no obstacle here: the register plan is made from the gate sequence itself)
    if (n >= kRegMinQubits && n <= kRegMaxQubits && c.passes.size() == 1) {
        // (QK_REG_LV = 12 at 10 or 11 qubits: four or two samples per workgroup of 256 threads, A/B runs)
        static const int lv_wide = std::getenv("QK_REG_LV") ? std::atoi(std::getenv("QK_REG_LV")) : 0;
        if (!(lv_wide > n && lv_wide <= kRegPackMaxWidth && plan_q(lv_wide, (size_t)160 * 1024))) plan_q(n, (size_t)160 * 1024);
    } else if (n >= kRegPackMinQubits && n < kRegMinQubits && c.passes.size() == 1 && c.passes[0].L == n) {
        // The width that keeps the most waves resident on a CU (LDS per workgroup against 3 waves per SIMD at 147 VGPRs);
        // among equals the narrowest workgroup, whose barriers cost least (9 qubits, GA genome: 3.9 ms at L = 10 with five
        // waves per CU, 5.1 at L = 11 and 4.3 at L = 12 with four; 8 qubits ZZ r = 2: 2.5 / 2.5 / 2.8 ms with 6 / 6 / 4).
        // QK_REG_LV forces one (A/B runs).
        static const int lv_env = std::getenv("QK_REG_LV") ? std::atoi(std::getenv("QK_REG_LV")) : 0;
        bool done = false;
        if (lv_env >= kRegMinQubits && lv_env <= kRegPackMaxWidth) done = plan_q(lv_env, (size_t)160 * 1024);
        int best = 0, best_waves = 0;
        bool widest = false;
        for (int Lv = kRegMinQubits; Lv <= kRegPackMaxWidth && !done; ++Lv) {
            size_t lds = 0;
            if (!plan_q(Lv, (size_t)160 * 1024, &lds)) continue;
            // a first layer that is not all Hadamards (GA genomes) has its product-state tables built inside the kernel, a
            // wave per sample and a barrier behind it: single-wave workgroups (L = 10) keep that wave-local
            // (end of round 3, eight random genomes of two gate alphabets at 7 / 8 / 9 qubits, tools/sim_bench.py --genomes 4: L = 11
            // or 12 wins or loses 5-20 % genome by genome, with no sign that the plan's counts predict; L = 10 stays)
            if (((lds >> 40) & 1) && n >= 7) { best = Lv; break; }      // (6 qubits and below: 16+ samples per wave either way, the count decides)
            // Diagonal stages with a layer's worth of terms that stay per thread at the narrowest width (ZZ map with full
            // entanglement: 24 / 32 / 48 terms at 7 / 8 / 9 qubits): the widest workgroup that fits, whose tables per
            // register position are affordable and whose term records are read once for 8 to 32 samples -- 4 GiB of Psi,
            // L = 10 -> 12: 9 qubits 4.45 -> 3.32 ms, 8 qubits 3.64 -> 3.15, 7 qubits 4.86 -> 4.38.
            if (Lv == kRegMinQubits && ((lds >> 42) & 1) && n >= 7) widest = true;
            if (widest) { best = Lv; continue; }
            const int rb = ((lds >> 41) & 1) ? 3 : 4;
            const int waves = std::min(rb == 3 ? 16 : 12, (int)((size_t)160 * 1024 / (lds & (((size_t)1 << 40) - 1))) * ((1 << (Lv - rb)) / 64));
            if (waves > best_waves) { best_waves = waves; best = Lv; }
        }
        if (!done && best) plan_q(best, (size_t)160 * 1024);
    }
    return QK_OK;
}

}  // namespace qk
