/*
 * send_pass_model.c -- CPU model of the wave-parallel SEND passes of pcc_sim.hip (heavy_mi).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  The HIP send half sends one env's packets 256 at a time with
 * all 64 lanes (4 packets per lane = one Philox block) using closed forms instead of the per-packet
 * recurrence of Link.packet_enters_link (ns:66-84).  This file restates those passes lane by lane
 * with the SAME floating-point / integer operations the kernel uses and checks them, bit for bit,
 * against the plain per-packet recurrence:
 *   (1) on fuzzed link states (pcc_model_fuzz), and
 *   (2) on the MI start states of real episodes driven by the oracle (pcc_model_episodes), which
 *       also reports how many packets each regime carries.
 *
 * Regimes of one pass (wave-uniform choice):
 *   A  "always empty"  every packet finds the queue drained (send gap >= 1/bw and the first packet
 *      already sees it empty): latency = dl, every packet that is not a random loss is accepted.
 *   B  "backlogged, one binade"  the queue never empties and q stays inside one binade [2^e, 2^(e+1)).
 *      Then every quantity is a multiple of u = ulp(q), fl(1/bw + x) = x + R with R = 1/bw rounded to
 *      a multiple of u, so after j accepted packets the queue seen at time t is EXACTLY
 *          x = q0 + j R - (t - tu0)
 *      and "accepted" is a token bucket: packet k is accepted iff it is not a random loss and
 *      j(k) < N_k,  N_k = floor((X* - q0 + (t_k - tu0)) / R) + 1,  X* = maxq - R.
 *      b_k = N_k - j(k) obeys Lindley's recursion b' = max(b - m, 0) + a, a (max,+)-linear map, so
 *      all 256 decisions come from one parallel prefix scan.  Packets that break a precondition
 *      (queue empties, q leaves the binade) are detected per packet; the pass commits the prefix
 *      before the first such packet.
 *   S  serial: a few packets with the plain recurrence (episode start, binade changes, ties).
 */
#include "../../oracle/pcc_oracle.c"

#include <stdio.h>

typedef struct { double t1, lat; } rec_t;

typedef struct {
    double q, tu, t;     /* link state and the next SEND time */
    uint32_t sent;       /* packets of this MI sent so far (Philox stream index) */
    uint32_t na, nd;     /* accepted / dropped records appended */
} sstate_t;

typedef struct {
    uint64_t pass_a, pass_b, pass_s, pk_a, pk_b, pk_s, b_empty_commit, pass_c, pk_c;
    uint64_t why[8];
} mstats_t;

static inline uint32_t exp_bits(double x) {
    uint64_t b;
    memcpy(&b, &x, 8);
    return (uint32_t)(b >> 52) & 0x7FFu;
}

/* ns:66-84, 170-175: one SEND at time t; returns dropped */
static int link_send_ref(double t, int rnd, double dl, double maxq, double ebw, double *q, double *tu, rec_t *rec) {
    const double qcur = py_max0(*q - (t - *tu));
    const double lat0 = dl + qcur;
    const int full = ebw + qcur > maxq;
    const double grown = qcur + ebw;
    if (!rnd) { *q = full ? qcur : grown; *tu = t; }
    rec->t1 = t + lat0;
    rec->lat = lat0;
    return rnd || full;
}

/* the plain recurrence over a whole MI: what every path must reproduce */
static void serial_mi(sstate_t *s, double gap, double end, double dl, double maxq, double ebw, const uint8_t *loss,
                      rec_t *acc, rec_t *drp) {
    while (s->t < end) {
        rec_t r;
        const int dropped = link_send_ref(s->t, loss[s->sent], dl, maxq, ebw, &s->q, &s->tu, &r);
        if (dropped) drp[s->nd++] = r; else acc[s->na++] = r;
        s->t += gap;
        s->sent++;
    }
}

/* lanes of one pass: 64 = one wavefront (heavy_mi<.., 1>), 256 = a team of four wavefronts (heavy_mi<.., 4>: the same pass over
 * 1 024 positions, the prefix sums carried from wavefront to wavefront through LDS).  pcc_model_set_lanes picks it. */
#define MAX_LANES 1024
#define PER_LANE 4
static int g_lanes = 64;
#define LANES g_lanes
#define PASS (g_lanes * PER_LANE)
#define MAX_PASS (MAX_LANES * PER_LANE)
static int g_fuzz_straddle = 0;
void pcc_model_fuzz_straddle(int on) { g_fuzz_straddle = on; }
int pcc_model_set_lanes(int lanes) {
    if (lanes < 1 || lanes > MAX_LANES) return -1;
    g_lanes = lanes;
    return 0;
}

/* one MI by passes; mirrors heavy_mi in pcc_sim.hip */
static void model_mi(sstate_t *s, double gap, double end, double dl, double maxq, double ebw, const uint8_t *loss,
                     rec_t *acc, rec_t *drp, mstats_t *st) {
    uint32_t serial_len = 8;
    while (s->t < end) {
        const double t0 = s->t;
        const double t1s = t0 + gap, G = t1s - t0, t2s = t1s + gap;
        /* positions whose send time leaves the binade of t0 are not part of the pass (their t0 + k G would
         * not be exact); lim = min(end, top of the binade) */
        const double ttop = ldexp(1.0, (int)exp_bits(t0) - 1022);
        const double lim = end < ttop ? end : ttop;
        const double tend = t0 + (double)PASS * G;
        const int ok_t = (t2s - t1s == G) && (G > 0.0) && (t0 >= (PASS + 4.0) * gap) && (exp_bits(t0) == exp_bits(t2s));
        const uint32_t skip = s->sent & 3u;   /* packets of lane 0's Philox block that are already sent */
        int regime = 0;                        /* 0 = serial, 1 = A, 2 = B */
        int why = 0;
        const double D0 = t0 - s->tu;
        /* ---- regime B constants (wave-uniform) */
        double u = 0, R = 0;
        int64_t Q0i = 0, D0i = 0, Gi = 0, Ri = 0, Mi = 0;
        uint32_t e = 0;
        int maxq_above = 0, free_mode = 0;
        if (!ok_t) why = 1;
        if (ok_t) {
            const double x0 = s->q - D0;
            if (G >= ebw && !(x0 > 0.0)) regime = 1;
            else {
                e = exp_bits(s->q);
                const uint32_t eb = exp_bits(ebw);
                int ok = (s->q > 0.0) && e > 64 && e < 1100 && (s->tu + s->tu >= tend) && (x0 > 0.0) && (eb <= e) &&
                         exp_bits(s->tu) >= e && exp_bits(maxq) >= e;
                if (!ok) why = 2;
                if (ok) {
                    u = ldexp(1.0, (int)e - 1023 - 52);
                    const double inv_u = ldexp(1.0, -((int)e - 1023 - 52));
                    const double probe = ldexp(1.0, (int)e - 1023);
                    R = (eb == e) ? ebw : (probe + ebw) - probe;
                    const double err = ebw - R;
                    const int tie = fabs(err) == 0.5 * u;
                    /* everything in units of u must fit an int64 with room for 2^10 packets' worth */
                    const double span = (D0 + (double)PASS * G) * inv_u;
                    ok = ok && span < 4.0e18 && R > 0.0;
                    if (!ok) why = 3;
                    if (ok) {
                        Q0i = (int64_t)(s->q * inv_u);
                        D0i = (int64_t)(D0 * inv_u);
                        Gi = (int64_t)(G * inv_u);
                        Ri = (int64_t)(R * inv_u);
                        /* room in the queue, in packets (estimate): with >= PASS + 44 every packet of the pass
                         * that is not lost at random is accepted and the token arithmetic is not needed
                         * (maxq / u may not even fit an int64 then) */
                        const double room = ((maxq - R) - x0) / R;
                        free_mode = room >= (double)PASS + 44.0;
                        Mi = free_mode ? 0 : (int64_t)(maxq * inv_u);
                        if (tie && ((Q0i | D0i | Gi) & 1)) { ok = 0; why = 4; }
                        maxq_above = exp_bits(maxq) > e;
                        /* the first packet would already take q out of the binade: no point in trying */
                        const uint32_t es0 = exp_bits(x0 + R);
                        if (ok && (es0 < e || (es0 > e && maxq_above))) { ok = 0; why = 6; }
                    }
                }
                if (ok) regime = 2;
            }
        }
        /* ---- C: "full queue straddling a power of two".  maxq sits just above B = 2^E (maxq - 1/bw < B <= maxq), so the
         * full queue lives in TWO binades: values below B are multiples of v = ulp(B) / 2, values from B up multiples of
         * 2 v, and fl(qcur + 1/bw) rounds to whichever grid its result lands on.  In units of v, with 1/bw = (I + f) v,
         * 0 < f < 1, f != 1/2:  a result below B is n + I + cl (cl = [f > 1/2]); a result from B up is n + I rounded up
         * to even.  Against the constant increment R0 = I + cl the accept of packet j is off by c'_j in {-1, 0, +1}, and
         * c'_j follows from where the result lands and from the parity of the queue before (a two-state automaton).  The
         * pass takes the accept/drop decisions and the landing sides from the base trajectory (increment R0 everywhere),
         * which differs from the true one by at most j units of v after j accepts -- a position whose decision or
         * landing side is closer than that to its threshold ends the pass (flag) -- then runs the automaton over the
         * accepted packets and adds the accumulated corrections to the queue every packet sees.  */
        int regime_c = 0;
        if (ok_t && regime != 1) {
            const uint32_t eM = exp_bits(maxq), eq = exp_bits(s->q), eb = exp_bits(ebw);
            const double B = ldexp(1.0, (int)eM - 1023);
            const double x0 = s->q - D0;
            int ok = (s->q > 0.0) && eM > 66 && eM < 1100 && (eq == eM || eq + 1 == eM) && (maxq - 64.0 * ebw < B) && (x0 > 0.0) &&
                     (s->tu + s->tu >= tend) && exp_bits(s->tu) >= eM && eb + 1 <= eM - 1;
            if (ok) {
                const double v = ldexp(1.0, (int)eM - 1 - 1023 - 52), inv_v = ldexp(1.0, -((int)eM - 1 - 1023 - 52));
                const double probe = ldexp(1.0, (int)eM - 1 - 1023);
                const double R0 = (probe + ebw) - probe;          /* 1/bw on the grid of v */
                const double errv = ebw - R0;
                const double span = (D0 + (double)PASS * G) * inv_v;
                ok = span < 4.0e18 && R0 > 0.0 && errv != 0.0 && fabs(errv) != 0.5 * v;
                if (ok) {
                    const int64_t Q0v = (int64_t)(s->q * inv_v), D0v = (int64_t)(D0 * inv_v), Gv = (int64_t)(G * inv_v);
                    const int64_t Rv = (int64_t)(R0 * inv_v), Mv = (int64_t)(maxq * inv_v), Bv = (int64_t)(B * inv_v);
                    const int64_t cl = errv < 0.0 ? 1 : 0, Iv = Rv - cl;
                    if (!(Gv < Rv)) ok = 0;                      /* the sender is not faster than the link: not this regime */
                    if (ok) {
                        const int64_t C = (Mv - Rv) - Q0v + D0v;
                        uint8_t m_k[MAX_PASS], ex_k[MAX_PASS], acc_k[MAX_PASS], flag_k[MAX_PASS], up_k[MAX_PASS];
                        int64_t xi_k[MAX_PASS];
                        /* decisions of the base trajectory: the token bucket of regime B, serially here (the kernel: the scan) */
                        int64_t j = 0;
                        for (uint32_t p = 0; p < (uint32_t)PASS; p++) {
                            const int32_t k = (int32_t)p - (int32_t)skip;
                            const double tk = t0 + (double)(k < 0 ? 0 : k) * G;
                            ex_k[p] = k >= 0 && tk < lim;
                            m_k[p] = ex_k[p] && !loss[s->sent + (k < 0 ? 0 : k)];
                            const int64_t kk = k < 0 ? 0 : k;
                            const int64_t xi = Q0v + j * Rv - D0v - kk * Gv;   /* base queue this packet sees */
                            xi_k[p] = xi;
                            /* tokens: accepted iff j < floor((C + k G) / R) + 1  <=>  xi + R <= M */
                            const int a = m_k[p] && (xi + Rv <= Mv);
                            acc_k[p] = (uint8_t)a;
                            int f = 0;
                            if (m_k[p]) {
                                const int64_t slack = (xi + Rv) - Mv, land = (xi + Iv) - Bv;
                                const int64_t mar = j + 2;
                                if (slack >= -mar && slack <= mar) f = 1;                      /* decision too close to call */
                                if (a && land >= -mar && land <= mar) f = 1;                   /* landing side too close to call */
                                if (xi - mar <= 0) f = 1;                                      /* the queue runs empty */
                                if (xi + Rv - mar < Bv / 2 + 2) f = 1;                         /* below the lower binade */
                            }
                            flag_k[p] = (uint8_t)f;
                            up_k[p] = (uint8_t)(a && (xi + Iv >= Bv));
                            if (a && k >= 0) j++;
                        }
                        /* commit the prefix before the first flagged packet: the automaton over its accepted packets */
                        uint32_t ncommit = 0;
                        int64_t P = Q0v & 1, Cacc = 0;
                        double last_q = 0, last_t = 0;
                        int any = 0;
                        for (uint32_t p = skip; p < (uint32_t)PASS; p++) {
                            if (!ex_k[p] || flag_k[p]) break;
                            const uint32_t k = p - skip;
                            const double tk = t0 + (double)k * G;
                            const int64_t xt = xi_k[p] + Cacc;                                  /* the true queue this packet sees */
                            const double qc = py_max0((double)xt * v);
                            rec_t r;
                            r.lat = dl + qc;
                            r.t1 = tk + r.lat;
                            if (acc_k[p]) {
                                int64_t c;
                                if (up_k[p]) { c = (P + Iv) & 1; P = 0; }
                                else { c = cl; P ^= (Iv + cl) & 1; }
                                Cacc += c - cl;
                                acc[s->na++] = r;
                                last_q = (double)(xi_k[p] + Rv + Cacc) * v;                       /* = xt + I + c */
                                last_t = tk; any = 1;
                            } else {
                                drp[s->nd++] = r;
                                if (m_k[p]) { last_q = (double)xt * v; last_t = tk; any = 1; }
                            }
                            ncommit++;
                        }
                        if (ncommit) {
                            if (any) { s->q = last_q; s->tu = last_t; }
                            s->t = (t0 + (double)(ncommit - 1) * G) + gap;
                            s->sent += ncommit;
                            st->pass_c++; st->pk_c += ncommit;
                            serial_len = 8;
                            regime_c = 1;
                        }
                    }
                }
            }
        }
        if (regime_c) continue;
        if (regime == 1) {
            /* ---- A: latency dl, accepted unless lost at random */
            uint32_t n = 0;
            double last_t = 0;
            int any = 0;
            for (uint32_t p = skip; p < PASS; p++) {
                const uint32_t k = p - skip;
                const double tk = t0 + (double)k * G;
                if (!(tk < lim)) break;
                const int rnd = loss[s->sent + k];
                rec_t r;
                r.lat = dl + 0.0;
                r.t1 = tk + r.lat;
                if (rnd) drp[s->nd++] = r; else { acc[s->na++] = r; last_t = tk; any = 1; }
                n++;
            }
            if (any) { s->q = ebw + 0.0; s->tu = last_t; }
            if (n) s->t = (t0 + (double)(n - 1) * G) + gap;
            s->sent += n;
            st->pass_a++; st->pk_a += n;
            serial_len = 8;
            continue;
        }
        if (regime == 2) {
            /* ---- B: token-bucket scan.  Lane l owns pass positions 4l..4l+3; packet k = position - skip
             * (positions before `skip` belong to packets the lane rounds already sent: they do not exist). */
            const int64_t C = (Mi - Ri) - Q0i + D0i;   /* A_k = C + k Gi;  N_k = floor((A_k + Ri) / Ri) >= 0 */
            const int over = !free_mode && Gi < Ri;     /* overdriven and close to full: the token scan decides */
            uint8_t m_k[MAX_PASS], ex_k[MAX_PASS], acc_k[MAX_PASS], flag_k[MAX_PASS];
            double x_k[MAX_PASS];
            for (uint32_t p = 0; p < PASS; p++) {
                const int32_t k = (int32_t)p - (int32_t)skip;
                const double tk = t0 + (double)(k < 0 ? 0 : k) * G;
                ex_k[p] = k >= 0 && tk < lim;
                m_k[p] = ex_k[p] && !loss[s->sent + (k < 0 ? 0 : k)];
            }
            /* phase 1 (overdriven only), per lane: tokens at the lane's first packet by one division
             * (double estimate + exact integer correction), token arrivals a of its packets, and the
             * lane's composite Lindley map b -> max(b + S, C) */
            int32_t S_l[MAX_LANES], C_l[MAX_LANES], N_l[MAX_LANES], a_k[MAX_PASS];
            for (int l = 0; l < LANES && over; l++) {
                int32_t k0 = 4 * l - (int32_t)skip;
                if (k0 < 0) k0 = 0;
                const int64_t num = C + Ri + (int64_t)k0 * Gi;   /* >= 0 */
                int64_t N = (int64_t)((double)num / (double)Ri);
                int64_t rem = num - N * Ri;
                if (rem < 0) { N--; rem += Ri; }
                if (rem >= Ri) { N++; rem -= Ri; }
                N_l[l] = (int32_t)N;
                int32_t Ssum = 0, Cmax = INT32_MIN / 2;
                for (int i = 0; i < PER_LANE; i++) {
                    const uint32_t p = 4 * l + i;
                    int32_t a = 0;
                    if ((int32_t)p >= (int32_t)skip) {   /* arrivals run on past the MI end: harmless */
                        rem += Gi;
                        if (rem >= Ri) { rem -= Ri; a = 1; }
                    }
                    a_k[p] = a;
                    const int32_t sft = a - (int32_t)m_k[p];
                    Cmax = (Cmax + sft > a) ? Cmax + sft : a;   /* compose this packet's map after the earlier ones */
                    Ssum += sft;
                }
                S_l[l] = Ssum; C_l[l] = Cmax;
            }
            /* phase 2: exclusive scan of the lane composites (the kernel: 6 DPP steps) */
            int32_t bin_l[MAX_LANES], jin_l[MAX_LANES];
            {
                int32_t preS = 0, preC = INT32_MIN / 2, jrun = 0;
                const int32_t b0 = over ? N_l[0] : ((free_mode || C >= 0) ? 1 : 0);
                for (int l = 0; l < LANES; l++) {
                    if (over) {
                        const int32_t viaS = b0 + preS;
                        bin_l[l] = viaS > preC ? viaS : preC;
                        const int32_t nC = (preC + S_l[l] > C_l[l]) ? preC + S_l[l] : C_l[l];
                        preS += S_l[l]; preC = nC;
                    } else {
                        /* underdriven: every packet after the first has a token; accepted = not lost.
                         * accepted packets before the lane = prefix popcount (the kernel: ballots) */
                        bin_l[l] = b0;
                        jin_l[l] = jrun;
                        for (int i = 0; i < PER_LANE; i++) {
                            const uint32_t p = 4 * l + i;
                            const int first = p == skip;
                            jrun += m_k[p] && (!first || b0 > 0);
                        }
                    }
                }
            }
            /* phase 3, per lane: decisions, the exact queue each packet sees, precondition flags */
            for (int l = 0; l < LANES; l++) {
                int32_t b = bin_l[l], N = over ? N_l[l] : 0, j = over ? N - b : jin_l[l];
                int32_t k0 = 4 * l - (int32_t)skip;
                if (k0 < 0) k0 = 0;
                /* exact base: x = (Q0 + j R - D0 - k0 G) u in integers, then one exact conversion */
                const int64_t xi = Q0i + (int64_t)j * Ri - D0i - (int64_t)k0 * Gi;
                double x = (double)xi * u;
                for (int i = 0; i < PER_LANE; i++) {
                    const uint32_t p = 4 * l + i;
                    const int exists_k = (int32_t)p >= (int32_t)skip;
                    int a;
                    if (over) a = m_k[p] && b > 0;
                    else a = m_k[p] && (p != skip || b > 0);
                    acc_k[p] = (uint8_t)a;
                    x_k[p] = x;
                    int f = 0;
                    const double sx = x + R;   /* the queue after this packet if it is accepted */
                    if (m_k[p]) {
                        const uint32_t es = exp_bits(sx);
                        f = !(x > 0.0) || es < e || (es > e && maxq_above);
                    }
                    flag_k[p] = (uint8_t)f;
                    if (exists_k) {
                        if (over) b = (b - (int32_t)m_k[p] > 0 ? b - (int32_t)m_k[p] : 0) + a_k[p];
                        x = (a ? sx : x) - G;   /* exact: multiples of u below 2^(e+1) */
                    }
                }
            }
            /* ---- commit the prefix before the first flagged packet */
            uint32_t ncommit = 0;
            double last_q = 0, last_t = 0;
            int any = 0;
            for (uint32_t p = skip; p < PASS; p++) {
                if (!ex_k[p] || flag_k[p]) break;
                const uint32_t k = p - skip;
                const double tk = t0 + (double)k * G;
                const double qc = py_max0(x_k[p]);
                rec_t r;
                r.lat = dl + qc;
                r.t1 = tk + r.lat;
                if (acc_k[p]) acc[s->na++] = r; else drp[s->nd++] = r;
                if (m_k[p]) { last_q = acc_k[p] ? x_k[p] + R : x_k[p]; last_t = tk; any = 1; }
                ncommit++;
            }
            if (ncommit) {
                if (any) { s->q = last_q; s->tu = last_t; }
                s->t = (t0 + (double)(ncommit - 1) * G) + gap;
                s->sent += ncommit;
                st->pass_b++; st->pk_b += ncommit;
                serial_len = 8;
                continue;
            }
            st->b_empty_commit++;
            why = 5;
        }
        /* ---- S: the plain recurrence for a few packets */
        st->why[why]++;
        uint32_t n = 0;
        while (n < serial_len && s->t < end) {
            rec_t r;
            const int dropped = link_send_ref(s->t, loss[s->sent], dl, maxq, ebw, &s->q, &s->tu, &r);
            if (dropped) drp[s->nd++] = r; else acc[s->na++] = r;
            s->t += gap;
            s->sent++;
            n++;
        }
        st->pass_s++; st->pk_s += n;
        if (serial_len < 64) serial_len *= 2;
    }
}

static int compare_mi(const sstate_t *a, const sstate_t *b, const rec_t *aa, const rec_t *ad, const rec_t *ba, const rec_t *bd) {
    if (a->sent != b->sent || a->na != b->na || a->nd != b->nd) return 1;
    if (memcmp(&a->q, &b->q, 8) || memcmp(&a->tu, &b->tu, 8) || memcmp(&a->t, &b->t, 8)) return 2;
    if (memcmp(aa, ba, sizeof(rec_t) * a->na)) return 3;
    if (memcmp(ad, bd, sizeof(rec_t) * a->nd)) return 4;
    return 0;
}

/* xorshift for the fuzzers (not the simulator's RNG) */
static uint64_t fz_state = 88172645463325252ull;
static uint64_t fz_next(void) { fz_state ^= fz_state << 13; fz_state ^= fz_state >> 7; fz_state ^= fz_state << 17; return fz_state; }
static double fz_unit(void) { return (double)(fz_next() >> 11) * (1.0 / 9007199254740992.0); }

#define MAXPK 70000

/* Fuzz: random links and mid-episode states, the MI sent twice.  Returns the number of mismatches. */
long pcc_model_fuzz(long n_cases, uint64_t seed, uint64_t *stats_out /* [16] */) {
    fz_state = seed ? seed : 1;
    mstats_t st;
    memset(&st, 0, sizeof st);
    long bad = 0;
    rec_t *aa = malloc(sizeof(rec_t) * MAXPK), *ad = malloc(sizeof(rec_t) * MAXPK);
    rec_t *ba = malloc(sizeof(rec_t) * MAXPK), *bd = malloc(sizeof(rec_t) * MAXPK);
    uint8_t *loss = malloc(MAXPK + 8);
    for (long c = 0; c < n_cases; c++) {
        const double bw = 100.0 + 400.0 * fz_unit();
        const double dl = 0.05 + 0.45 * fz_unit();
        double queue = (double)(1 + (long)exp(8.0 * fz_unit()));
        const int straddle = g_fuzz_straddle && (fz_next() & 1);
        if (straddle) {   /* a queue limit just above a power of two: the full queue straddles it */
            const double B = ldexp(1.0, (int)(fz_next() % 7) - 2);       /* 0.25 .. 16 s */
            queue = ceil(B * bw + fz_unit() * 0.9);
            if (queue < 2.0) queue = 2.0;
        }
        const double lr = (fz_next() & 7) == 0 ? 0.0 : 0.05 * fz_unit();
        const double maxq = queue / bw, ebw = 1.0 / bw;
        double rate = (fz_next() & 3) == 0 ? 1000.0 : 40.0 + 960.0 * fz_unit();
        if (straddle) rate = bw * (1.02 + 0.9 * fz_unit());            /* overdriven: the queue fills and stays full */
        if (rate > 1000.0) rate = 1000.0;
        if ((fz_next() & 7) == 0) rate = bw * (0.98 + 0.04 * fz_unit());
        const double gap = 1.0 / rate;
        /* a state as an episode would leave it: run the plain recurrence for a random while first */
        sstate_t s0;
        memset(&s0, 0, sizeof s0);
        s0.t = gap;
        const double warm_rate = straddle ? rate : 40.0 + 960.0 * fz_unit();
        const double warm_end = straddle ? 4.0 * maxq + 2.0 + fz_unit() * 100.0 : fz_unit() * fz_unit() * 400.0;
        double q = 0, tu = 0, t = 1.0 / warm_rate;
        long guard = 0;
        while (t < warm_end && guard++ < 2000000) {
            rec_t r;
            (void)link_send_ref(t, fz_unit() < lr, dl, maxq, ebw, &q, &tu, &r);
            t += 1.0 / warm_rate;
        }
        s0.q = q; s0.tu = tu; s0.t = t;
        s0.sent = (uint32_t)(fz_next() & 3);   /* Philox block alignment of a take-over */
        double dur = (0.1 + 30.0 * fz_unit() * fz_unit());
        if (dur * rate > 60000.0) dur = 60000.0 / rate;
        const double end = t + dur;
        for (long i = 0; i < MAXPK; i++) loss[i] = fz_unit() < lr;
        sstate_t sa = s0, sb = s0;
        serial_mi(&sa, gap, end, dl, maxq, ebw, loss, aa, ad);
        model_mi(&sb, gap, end, dl, maxq, ebw, loss, ba, bd, &st);
        const int rc = compare_mi(&sa, &sb, aa, ad, ba, bd);
        if (rc) {
            if (bad < 5)
                fprintf(stderr, "fuzz case %ld mismatch rc=%d bw=%.17g dl=%.17g queue=%g lr=%g rate=%.17g q0=%.17g tu0=%.17g t0=%.17g end=%.17g sent %u/%u na %u/%u\n",
                        c, rc, bw, dl, queue, lr, rate, s0.q, s0.tu, s0.t, end, sa.sent, sb.sent, sa.na, sb.na);
            bad++;
        }
    }
    if (stats_out) {
        stats_out[0] = st.pass_a; stats_out[1] = st.pass_b; stats_out[2] = st.pass_s;
        stats_out[3] = st.pk_a; stats_out[4] = st.pk_b; stats_out[5] = st.pk_s; stats_out[6] = st.b_empty_commit;
        for (int i = 0; i < 8; i++) stats_out[7 + i] = st.why[i];
        stats_out[14] = st.pass_c; stats_out[15] = st.pk_c;
    }
    free(aa); free(ad); free(ba); free(bd); free(loss);
    return bad;
}

static double pending_send_time(const env_t *e) {
    for (long i = 0; i < e->heap_n; i++)
        if (e->heap[i].type == EV_SEND && e->heap[i].hop == 0) return e->heap[i].t;
    return -1.0;
}

/* Real MI start states: the oracle runs n_envs episodes of n_steps steps (Philox uniforms, bench
 * ranges, U(-1,1) actions); before every step the SEND stream of the coming MI is sent twice from
 * the oracle's state -- plain recurrence vs passes -- and compared.  min_packets: only MIs with at
 * least that many packets go through the pass model (the kernel's heavy threshold). */
long pcc_model_episodes(int n_envs, int n_steps, uint64_t seed, uint32_t gid_base, uint32_t min_packets,
                        uint64_t *stats_out /* [16] */, uint64_t *hist_out /* [32]: MIs by log2(packets) */) {
    mstats_t st;
    memset(&st, 0, sizeof st);
    long bad = 0;
    rec_t *aa = malloc(sizeof(rec_t) * MAXPK), *ad = malloc(sizeof(rec_t) * MAXPK);
    rec_t *ba = malloc(sizeof(rec_t) * MAXPK), *bd = malloc(sizeof(rec_t) * MAXPK);
    uint8_t *loss = malloc(MAXPK + 8);
    const int fid[3] = {7, 10, 11};
    fz_state = seed * 2654435761u + 12345;
    uint64_t total_pk = 0, heavy_pk = 0, heavy_mis = 0, total_mis = 0;
    for (int b = 0; b < n_envs; b++) {
        env_t *e = pcc_oracle_create(1, 10, fid, 3, MEAN_NUMPY);
        pcc_oracle_rng_philox(e, seed, gid_base + (uint32_t)b);
        pcc_oracle_reset(e, NULL);
        for (int tstep = 0; tstep < n_steps; tstep++) {
            double act = 2.0 * fz_unit() - 1.0;
            /* the state the send half starts from, with the action applied (ns:235-241) */
            sender_t tmp = e->snd[0];
            sender_apply_rate_delta(&tmp, act, 0.025);
            const double gap = 1.0 / tmp.rate;
            sstate_t s0;
            memset(&s0, 0, sizeof s0);
            s0.q = e->links[0].queue_delay; s0.tu = e->links[0].queue_delay_update_time;
            s0.t = pending_send_time(e);
            const double end = e->cur_time + e->run_dur;
            const double lr = e->links[0].lr, dl = e->links[0].dl, maxq = e->links[0].max_queue_delay;
            const double ebw = 1.0 / e->links[0].bw;
            const double ahead = s0.t < end ? (end - s0.t) / gap + 4.0 : 0.0;
            if (ahead < MAXPK - 8) {
                const uint32_t np = (uint32_t)ahead + 4;
                for (uint32_t j = 0; j < np; j += 4) {
                    uint32_t ctr[4] = {j >> 2, (uint32_t)e->mi_index, (uint32_t)e->episode, e->ph_gid}, out[4];
                    philox4x32_10(ctr, e->ph_key, out);
                    for (int x = 0; x < 4; x++) loss[j + x] = (double)out[x] * (1.0 / 4294967296.0) < lr;
                }
                sstate_t sa = s0, sb = s0;
                serial_mi(&sa, gap, end, dl, maxq, ebw, loss, aa, ad);
                total_pk += sa.sent; total_mis++;
                int lg = 0;
                while ((1u << lg) < sa.sent + 1u && lg < 31) lg++;
                if (hist_out) hist_out[lg]++;
                if (sa.sent >= min_packets) {
                    heavy_pk += sa.sent; heavy_mis++;
                    const mstats_t before = st;
                    model_mi(&sb, gap, end, dl, maxq, ebw, loss, ba, bd, &st);
                    if (getenv("PCC_MODEL_VERBOSE") && sa.sent >= 800) {  /* which MIs need many passes, and why */
                        const uint64_t passes = (st.pass_a - before.pass_a) + (st.pass_b - before.pass_b) + (st.pass_s - before.pass_s);
                        if (sa.sent / (passes ? passes : 1) < (uint64_t)atoi(getenv("PCC_MODEL_VERBOSE")))
                            fprintf(stderr, "env %d step %d sent %u passes A %llu B %llu S %llu (pk S %llu) Q %.0f rate/bw %.3f q/maxq %.6f "
                                    "maxq %.17g ebw %.17g q %.17g t %.6f why %llu %llu %llu %llu %llu %llu\n", b, tstep, sa.sent,
                                    (unsigned long long)(st.pass_a - before.pass_a), (unsigned long long)(st.pass_b - before.pass_b),
                                    (unsigned long long)(st.pass_s - before.pass_s), (unsigned long long)(st.pk_s - before.pk_s),
                                    maxq / ebw, ebw / gap, s0.q / maxq, maxq, ebw, s0.q, s0.t,
                                    (unsigned long long)(st.why[0] - before.why[0]), (unsigned long long)(st.why[1] - before.why[1]),
                                    (unsigned long long)(st.why[2] - before.why[2]), (unsigned long long)(st.why[3] - before.why[3]),
                                    (unsigned long long)(st.why[4] - before.why[4]), (unsigned long long)(st.why[5] - before.why[5]));
                    }
                    const int rc = compare_mi(&sa, &sb, aa, ad, ba, bd);
                    if (rc) {
                        if (bad < 5) fprintf(stderr, "episode env %d step %d mismatch rc=%d sent %u/%u\n", b, tstep, rc, sa.sent, sb.sent);
                        bad++;
                    }
                }
            }
            double obs[30], rew[1];
            pcc_oracle_step(e, &act, 0.025, obs, rew, NULL);
        }
        pcc_oracle_destroy(e);
    }
    if (stats_out) {
        stats_out[0] = st.pass_a; stats_out[1] = st.pass_b; stats_out[2] = st.pass_s;
        stats_out[3] = st.pk_a; stats_out[4] = st.pk_b; stats_out[5] = st.pk_s; stats_out[6] = st.b_empty_commit;
        for (int i = 0; i < 6; i++) stats_out[7 + i] = st.why[i];
        stats_out[13] = total_pk; stats_out[14] = heavy_pk; stats_out[15] = heavy_mis;
    }
    (void)total_mis;
    free(aa); free(ad); free(ba); free(bd); free(loss);
    return bad;
}
