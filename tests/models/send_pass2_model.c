/* CPU model of the two-sender wave path's token pass (pcc-rl_amd/csrc/pcc_sim.hip: heavy_mi2), with the kernel's own
 * integer / floating-point operations, against the plain merged recurrence (ns:66-84, 155-178 for two senders sharing the
 * link; events of equal time: sender 0 first).  Test infrastructure (tests/test_send_pass_model.py): a mismatch is a
 * counter-example for the pass's preconditions, found without a GPU.
 *
 *   gcc -O2 -fPIC -shared -ffp-contract=off send_pass2_model.c -o libsend_pass2_model.so -lm
 *
 * The pass covers `lanes * per_lane` merged positions (the kernel: 64 x 1); per_lane = 4 is the variant with one Philox
 * block per lane.  Whatever the token pass does not commit is sent by the plain recurrence, 64 packets at a time (the
 * kernel's accept chain / serial pass, which have no preconditions).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { double t1, lat; } rec_t;
typedef struct {
    double q, tu, t[2];
    uint32_t a[2], d[2];
} st2_t;

static uint32_t exp_bits(double x) {
    uint64_t b;
    memcpy(&b, &x, 8);
    return (uint32_t)((b >> 52) & 0x7FFu);
}
static double pow2(int e) {
    uint64_t b = (uint64_t)(e + 1023) << 52;
    double x;
    memcpy(&x, &b, 8);
    return x;
}
static double max0(double x) { return x > 0.0 ? x : 0.0; }

/* ns:66-84 */
static int link_send(double t, int rnd, double dl, double maxq, double ebw, double *q, double *tu, rec_t *rec) {
    const double qd = max0(*q - (t - *tu));
    rec->lat = dl + qd;
    rec->t1 = t + rec->lat;
    if (rnd) return 1;
    *q = qd;
    *tu = t;
    if (ebw + *q > maxq) return 1;
    *q += ebw;
    return 0;
}

/* up to `limit` packets of the merged stream by the plain recurrence; loss[k] = k-th packet of the interval is lost at random */
static uint32_t plain(st2_t *s, const double gap[2], double end, double dl, double maxq, double ebw, const uint8_t *loss,
                      uint32_t k0, uint32_t limit, rec_t *acc[2], rec_t *drp[2]) {
    uint32_t k = 0;
    while (k < limit) {
        const int sd = s->t[1] < s->t[0] ? 1 : 0;
        const double t = s->t[sd];
        if (!(t < end)) break;
        rec_t r;
        const int dropped = link_send(t, loss[k0 + k], dl, maxq, ebw, &s->q, &s->tu, &r);
        if (dropped) drp[sd][s->d[sd]++] = r;
        else acc[sd][s->a[sd]++] = r;
        s->t[sd] = t + gap[sd];
        k++;
    }
    return k;
}

static int g_lanes = 64, g_per_lane = 1;
int pcc_model2_set_shape(int lanes, int per_lane) {
    if (lanes < 1 || lanes > 1024 || per_lane < 1 || per_lane > 4) return -1;
    g_lanes = lanes;
    g_per_lane = per_lane;
    return 0;
}

#define MAXPOS 4096

/* one token pass; returns the packets it committed (0: preconditions not met / first packet flagged) */
static uint32_t token_pass(st2_t *st, const double gap[2], double end, double dl, double maxq, double ebw, const uint8_t *loss,
                           uint32_t k0, rec_t *acc[2], rec_t *drp[2], uint32_t *stopped_early) {
    const uint32_t kPass = (uint32_t)(g_lanes * g_per_lane);
    double G[2];
    int okb = 1;
    double tend_max = 0.0;
    for (int s = 0; s < 2; s++) {
        const double t0 = st->t[s], t1s = t0 + gap[s];
        G[s] = t1s - t0;
        const double t2s = t1s + gap[s], tend = t0 + (double)kPass * G[s];
        okb = okb && (t2s - t1s == G[s]) && (G[s] > 0.0) && (t0 >= ((double)kPass + 4.0) * gap[s]) && (exp_bits(t0) == exp_bits(tend));
        if (tend > tend_max) tend_max = tend;
    }
    const uint32_t e = exp_bits(st->q), eb = exp_bits(ebw);
    const double T0 = st->t[0] <= st->t[1] ? st->t[0] : st->t[1];
    const double x0 = st->q - (T0 - st->tu);
    okb = okb && (st->tu >= maxq) && (st->tu + st->tu >= tend_max) && (st->q > 0.0) && e > 64u && e < 1100u && (x0 > 0.0) && eb <= e &&
          exp_bits(st->tu) >= e && exp_bits(maxq) >= e;
    if (!okb) return 0;
    const double u = pow2((int)e - 1023 - 52), inv_u = pow2(-((int)e - 1023 - 52)), probe = pow2((int)e - 1023);
    const double R = (eb == e) ? ebw : (probe + ebw) - probe;
    const double err = ebw - R;
    const int tie = fabs(err) == 0.5 * u;
    if (!((tend_max - st->tu) * inv_u < 4.0e18 && R > 0.0)) return 0;
    const int64_t Q0i = (int64_t)(st->q * inv_u), Ri = (int64_t)(R * inv_u);
    int64_t Dsi[2], Gsi[2];
    for (int s = 0; s < 2; s++) {
        Dsi[s] = (int64_t)((st->t[s] - st->tu) * inv_u);
        Gsi[s] = (int64_t)(G[s] * inv_u);
    }
    const double room = ((maxq - R) - x0) / R;
    const int free_mode = room >= (double)kPass + 44.0;
    const int64_t Mi = free_mode ? 0 : (int64_t)(maxq * inv_u);
    if (tie && ((Q0i | Dsi[0] | Dsi[1] | Gsi[0] | Gsi[1]) & 1)) return 0;
    const int maxq_above = exp_bits(maxq) > e;

    /* the merged positions (the kernel finds them per lane by a merge-path search) */
    static double tk[MAXPOS];
    static int64_t Dp[MAXPOS];
    static int sd[MAXPOS], m[MAXPOS], ex[MAXPOS], N[MAXPOS + 1], a_[MAXPOS], accv[MAXPOS], flag[MAXPOS];
    uint32_t c[2] = {0, 0};
    for (uint32_t p = 0; p < kPass; p++) {
        const double A = st->t[0] + (double)c[0] * G[0], B = st->t[1] + (double)c[1] * G[1];
        const int is1 = !(A <= B);
        sd[p] = is1;
        tk[p] = is1 ? B : A;
        Dp[p] = Dsi[is1] + (int64_t)c[is1] * Gsi[is1];
        ex[p] = tk[p] < end;
        m[p] = ex[p] && !loss[k0 + p];
        c[is1]++;
    }
    /* accept decisions */
    if (free_mode) {
        for (uint32_t p = 0; p < kPass; p++) accv[p] = m[p];
    } else {
        for (uint32_t p = 0; p < kPass; p++) {
            const int64_t num = (Mi - Q0i) + Dp[p];
            int n = (int)((double)num * (1.0 / (double)Ri));
            int64_t rem = num - (int64_t)n * Ri;
            if (rem < 0) { n--; rem += Ri; }
            if (rem >= Ri) { n++; }
            N[p] = n;
        }
        N[kPass] = N[kPass - 1];   /* (the last lane's last position: no arrival behind it) */
        /* the kernel: per-lane composites + prefix scan; here the same Lindley recursion, position by position */
        int b = N[0];
        for (uint32_t p = 0; p < kPass; p++) {
            a_[p] = N[p + 1] - N[p];
            accv[p] = m[p] && b > 0;
            b = (b - m[p] > 0 ? b - m[p] : 0) + a_[p];
        }
    }
    int j = 0;
    uint32_t p_stop = kPass;
    static double xq[MAXPOS];
    for (uint32_t p = 0; p < kPass; p++) {
        const int64_t xi = Q0i + (int64_t)j * Ri - Dp[p];
        xq[p] = (double)xi * u;
        const double sx = xq[p] + R;
        const uint32_t es = exp_bits(sx);
        flag[p] = m[p] && (!(xq[p] > 0.0) || es < e || (es > e && maxq_above));
        if ((!ex[p] || flag[p]) && p < p_stop) p_stop = p;
        j += accv[p];
    }
    *stopped_early = p_stop < kPass && ex[p_stop < kPass ? p_stop : 0];
    if (!p_stop) return 0;
    uint32_t n[2] = {0, 0};
    for (uint32_t p = 0; p < p_stop; p++) {
        rec_t r;
        r.lat = dl + max0(xq[p]);
        r.t1 = tk[p] + r.lat;
        if (accv[p]) acc[sd[p]][st->a[sd[p]]++] = r;
        else drp[sd[p]][st->d[sd[p]]++] = r;
        if (m[p]) { st->q = accv[p] ? xq[p] + R : xq[p]; st->tu = tk[p]; }
        n[sd[p]]++;
    }
    st->t[0] = st->t[0] + (double)n[0] * G[0];
    st->t[1] = st->t[1] + (double)n[1] * G[1];
    return p_stop;
}

static uint64_t fz = 88172645463325252ull;
static uint64_t fz_next(void) { fz ^= fz << 13; fz ^= fz >> 7; fz ^= fz << 17; return fz; }
static double fz_unit(void) { return (double)(fz_next() >> 11) * (1.0 / 9007199254740992.0); }

#define MAXPK 40000

/* n_cases random link states and intervals; returns the number of mismatching cases.  stats: [0] token passes that
 * committed, [1] packets they committed, [2] packets sent by the plain recurrence, [3] passes stopped early by a flag */
long pcc_model2_fuzz(long n_cases, uint64_t seed, uint64_t *stats) {
    fz = seed ? seed : 1;
    long bad = 0;
    static rec_t ra[2][2][MAXPK], rd[2][2][MAXPK];
    static uint8_t loss[MAXPK + 8192];
    for (int k = 0; k < 4; k++) stats[k] = 0;
    for (long cs = 0; cs < n_cases; cs++) {
        const double bw = 100.0 + 400.0 * fz_unit();
        const double ebw = 1.0 / bw;
        double queue = 1.0 + floor(exp(8.0 * fz_unit()));
        if (fz_unit() < 0.35) {   /* queue limits around powers of two (in seconds) */
            const double B = pow2((int)(fz_next() % 8) - 3);
            queue = floor(B * bw + 6.0 * fz_unit() - 3.0);
            if (queue < 2.0) queue = 2.0;
        }
        const double maxq = queue / bw;
        const double dl = 0.05 + 0.45 * fz_unit();
        const double lr = fz_unit() < 0.25 ? 0.0 : 0.05 * fz_unit();
        const double load = fz_unit() < 0.7 ? 1.02 + 0.8 * fz_unit() : 0.4 + 1.2 * fz_unit();
        const double share = 0.2 + 0.6 * fz_unit();
        double rate[2] = {bw * load * share, bw * load * (1.0 - share)};
        for (int s = 0; s < 2; s++) { if (rate[s] < 40.0) rate[s] = 40.0; if (rate[s] > 1000.0) rate[s] = 1000.0; }
        const double gap[2] = {1.0 / rate[0], 1.0 / rate[1]};
        /* a clock somewhere in an episode (young episodes included), the queue anywhere up to full */
        const double now = fz_unit() < 0.2 ? 0.05 + 3.0 * fz_unit() : 3.0 + 400.0 * fz_unit();
        st2_t s0;
        s0.tu = now - ebw * fz_unit();
        s0.q = fz_unit() < 0.2 ? maxq * fz_unit() : (fz_unit() < 0.5 ? maxq - ebw * 3.0 * fz_unit() : maxq * (0.5 + 0.5 * fz_unit()));
        if (s0.q < 0.0) s0.q = 0.0;
        s0.t[0] = now + gap[0] * fz_unit();
        s0.t[1] = now + gap[1] * fz_unit();
        if (fz_unit() < 0.1) s0.t[1] = s0.t[0];   /* equal times: sender 0 first */
        s0.a[0] = s0.a[1] = s0.d[0] = s0.d[1] = 0;
        const double end = now + (0.05 + 2.0 * fz_unit());
        for (int k = 0; k < MAXPK + 8192; k++) loss[k] = fz_unit() < lr;
        if ((rate[0] + rate[1]) * (end - now) > (double)MAXPK - 100.0) continue;
        /* reference */
        st2_t r = s0;
        rec_t *racc[2] = {ra[0][0], ra[0][1]}, *rdrp[2] = {rd[0][0], rd[0][1]};
        uint32_t kr = 0;
        for (;;) {
            const uint32_t n = plain(&r, gap, end, dl, maxq, ebw, loss, kr, 1u << 30, racc, rdrp);
            kr += n;
            break;
        }
        /* model */
        st2_t md = s0;
        rec_t *macc[2] = {ra[1][0], ra[1][1]}, *mdrp[2] = {rd[1][0], rd[1][1]};
        uint32_t km = 0, chain_left = 0, guard = 0;
        while ((md.t[0] < md.t[1] ? md.t[0] : md.t[1]) < end) {
            if (++guard > 100000u) { km = 0xFFFFFFFFu; break; }
            uint32_t n = 0, early = 0;
            if (chain_left) chain_left--;
            else {
                n = token_pass(&md, gap, end, dl, maxq, ebw, loss, km, macc, mdrp, &early);
                if (early && n < (uint32_t)(g_per_lane > 1 ? 32 : 16)) chain_left = 2;
                if (n) { stats[0]++; stats[1] += n; stats[3] += early; }
            }
            if (!n) {
                n = plain(&md, gap, end, dl, maxq, ebw, loss, km, 64, macc, mdrp);
                stats[2] += n;
            }
            km += n;
        }
        int same = km == kr && md.q == r.q && md.tu == r.tu && md.t[0] == r.t[0] && md.t[1] == r.t[1];
        for (int s = 0; s < 2 && same; s++) {
            same = md.a[s] == r.a[s] && md.d[s] == r.d[s] && !memcmp(ra[0][s], ra[1][s], sizeof(rec_t) * r.a[s]) &&
                   !memcmp(rd[0][s], rd[1][s], sizeof(rec_t) * r.d[s]);
        }
        if (!same) bad++;
    }
    return bad;
}
