// lpp_tj_host.cpp -- host planner of the hole-major form of the one-orbital t-J model (no device code; rationale: lpp_tj_kernels.h), and
// lpp_tj_plan_stats: the plan expanded again on the host and compared with a CSR of the model entry by entry (the CPU suite runs it
// against the oracle's restatement of TjMultiOrb::setupHamiltonian).  Citations are relative to /root/reference/src.
#include <algorithm>
#include <cstring>

#include "../../include/lpp_engine.h"
#include "lpp_host.h"
#include "lpp_tj.h"

namespace lpp {

namespace {
uint64_t binom_t(int n, int k)
{
	if (k < 0 || k > n) return 0;
	uint64_t r = 1;
	for (int i = 1; i <= k; i++) r = r * (uint64_t)(n - k + i) / (uint64_t)i;
	return r;
}
// all nbits-bit words with k set bits, ascending (the loop of BasisOneSpin.h:53-61 / BasisTjMultiOrbLanczos.h:323-352)
std::vector<uint32_t> words_of_t(int nbits, int k)
{
	std::vector<uint32_t> out;
	if (k == 0) {
		out.push_back(0);
		return out;
	}
	if (k > nbits) return out;
	uint32_t w = (k >= 32) ? ~0u : ((1u << k) - 1u);
	const uint64_t limit = 1ull << nbits;
	while ((uint64_t)w < limit) {
		out.push_back(w);
		const uint32_t c = w & (0u - w), r = w + c; // Gosper's next word with the same popcount
		if (r == 0) break;
		w = (((r ^ w) >> 2) / c) | r;
	}
	return out;
}
} // namespace

void tj_plan(const TjModel& M, TjPlan& P, bool* ok, std::string* why)
{
	*ok = false;
	P = TjPlan();
	const int L = M.L, nup = M.nup, ndown = M.ndown, Lo = nup + ndown, nholes = L - Lo;
	if (L < 1 || L > 31 || nholes < 0 || Lo < 2 || Lo > 2 * kTjMaxHalf || nup < 1 || ndown < 1) {
		if (why) *why = "sector outside the form's limits";
		return;
	}
	const uint64_t ns64 = binom_t(Lo, nup), nblk64 = binom_t(L, nholes);
	if (ns64 * nblk64 >= ((uint64_t)1 << 31) || nblk64 > (1u << 20)) {
		if (why) *why = "too many states for 32-bit positions";
		return;
	}
	const int ns = (int)ns64, nblk = (int)nblk64;
	const int lb = Lo / 2, hb = Lo - lb;
	P.Lo = Lo;
	P.lb = lb;
	P.hb = hb;
	P.ns = ns;
	P.nblk = nblk;
	// ---- spin patterns and their ranking -----------------------------------------------------------------------------------------
	P.pat = words_of_t(Lo, nup);
	P.hi_base.assign((size_t)1 << hb, 0);
	P.lo_rank.assign((size_t)1 << lb, 0);
	{
		int64_t run = 0;
		for (uint32_t h = 0; h < (1u << hb); h++) {
			P.hi_base[h] = (int32_t)std::min<int64_t>(run, ns - 1);
			run += (int64_t)binom_t(lb, nup - __builtin_popcount(h));
		}
		std::vector<int> seen((size_t)lb + 1, 0);
		for (uint32_t l = 0; l < (1u << lb); l++) P.lo_rank[l] = (uint16_t)seen[(size_t)__builtin_popcount(l)]++;
	}
	// ---- work items: runs of whole segments (patterns sharing the bits above the low kbits), at most kTjWindow patterns ----------------
	int kbits = 1;
	for (int k = 1; k <= Lo; k++) {
		uint64_t longest = 0;
		for (int m = std::max(0, nup - (Lo - k)); m <= std::min(k, nup); m++) longest = std::max(longest, binom_t(k, m));
		if (longest <= (uint64_t)kTjWindow) kbits = k;
	}
	P.kbits = kbits;
	{
		int64_t r0 = 0;
		TjItem cur { 0, 0 };
		for (uint32_t t = 0; t < (1u << (Lo - kbits)); t++) {
			const int len = (int)binom_t(kbits, nup - __builtin_popcount(t));
			if (len == 0) continue;
			if (cur.len > 0 && cur.len + len > kTjWindow) {
				P.items.push_back(cur);
				cur = TjItem { (int32_t)r0, 0 };
			}
			if (cur.len == 0) cur.r0 = (int32_t)r0;
			cur.len += len;
			r0 += len;
		}
		if (cur.len > 0) P.items.push_back(cur);
		if (r0 != ns || (int)P.pat.size() != ns) {
			if (why) *why = "segments do not add up";
			return;
		}
	}
	// ---- hole configurations: bonds among the occupied sites, moves of an electron onto a neighbouring hole ------------------------
	P.holes = words_of_t(L, nholes);
	if ((int)P.holes.size() != nblk) {
		if (why) *why = "block count";
		return;
	}
	auto block_of = [&](uint32_t hm) -> int { return (int)(std::lower_bound(P.holes.begin(), P.holes.end(), hm) - P.holes.begin()); };
	P.blocks.resize((size_t)nblk);
	for (int b = 0; b < nblk; b++) {
		const uint32_t hm = P.holes[(size_t)b];
		int pos[32];
		int np = 0;
		for (int i = 0; i < L; i++) pos[i] = ((hm >> i) & 1u) ? -1 : np++;
		TjBlock B {};
		B.x_first = (int32_t)P.pairs.size();
		B.h_first = (int32_t)P.hops.size();
		for (int i = 0; i < L; i++)
			for (int j = i + 1; j < L; j++) { // the reference visits j >= i only (TjMultiOrb.h:666, 725)
				const double jv = M.jpm[(size_t)i * L + j];
				if (jv != 0 && pos[i] >= 0 && pos[j] >= 0) {
					TjPair pr;
					pr.mask = (1u << pos[i]) | (1u << pos[j]);
					pr.pad = 0;
					const double h = jv * 0.5; // :736
					pr.v = ((pos[j] - pos[i]) & 1) ? -h : h; // signSplusSminus (:772-783): the electrons on the sites [i, j) of the bra
					P.pairs.push_back(pr);
				}
				const double hr = M.hop_re[(size_t)i * L + j], hi = M.has_im ? M.hop_im[(size_t)i * L + j] : 0.0;
				if (hr == 0 && hi == 0) continue;
				if ((pos[i] >= 0) == (pos[j] >= 0)) continue; // one electron, one hole (:673, :683: the guards against double occupancy)
				int between = 0;
				for (int c = i + 1; c < j; c++) between += pos[c] >= 0 ? 1 : 0;
				TjHop hp;
				hp.vr = hr;
				hp.vi = hi;
				hp.m = (uint8_t)between;
				hp.pad = 0;
				if (pos[i] >= 0) { // the electron at i moves up to the hole at j
					hp.dir = 0;
					hp.lo = (uint8_t)pos[i];
					hp.dst = block_of((hm & ~(1u << j)) | (1u << i));
				} else { // the electron at j moves down to the hole at i
					hp.dir = 1;
					hp.lo = (uint8_t)(pos[j] - between);
					hp.dst = block_of((hm & ~(1u << i)) | (1u << j));
				}
				if (hi != 0) P.cplx_hops = true;
				P.hops.push_back(hp);
			}
		const int nxb = (int)P.pairs.size() - B.x_first, nhb = (int)P.hops.size() - B.h_first;
		if (nxb > kTjMaxPairs || nhb > kTjMaxHops) {
			if (why) *why = "more bonds / moves in one hole configuration than the kernel's lists hold";
			return;
		}
		// bonds among the low kbits positions first: their flips stay inside a segment (LDS reads of the item's window)
		const auto low_end = std::stable_partition(P.pairs.begin() + B.x_first, P.pairs.end(), [&](const TjPair& q) { return q.mask < (1u << kbits); });
		B.nx = (int16_t)nxb;
		B.nxl = (int16_t)(low_end - (P.pairs.begin() + B.x_first));
		B.nh = (int16_t)nhb;
		P.blocks[(size_t)b] = B;
	}
	*ok = true;
}

} // namespace lpp

using namespace lpp;

// The plan of the model, expanded on the host exactly as k_tj_apply walks it (flip of an antiparallel pair; rotation of the bits between
// an electron and a hole, sign of the same-species electrons between), and compared with the off-diagonal entries of a CSR of the same model
// in the reference's basis order (BasisTjMultiOrbLanczos.h:29-42: sorted (down << L) | up words) -- columns equal, values bit-identical.
// out[0] = 1 if the plan applies and reproduces every row, out[1..9] = hole configurations, spin patterns, low positions of a segment,
// work items, bonds, moves, bonds among the low positions, most bonds / moves of one configuration.
extern "C" lpp_status lpp_tj_plan_stats(int32_t L, int32_t nup, int32_t ndown, const double* hop_re, const double* hop_im, const double* jpm, int64_t nrows,
                                        const int64_t* rowptr, const int32_t* colind, const void* values, int32_t is_complex, int64_t* out)
{
	if (!hop_re || !jpm || !out || L < 1 || L > 31 || nup < 0 || ndown < 0 || nup + ndown > L) return fail(LPP_ERR_INVALID, "lpp_tj_plan_stats: bad argument");
	for (int k = 0; k < 10; k++) out[k] = 0;
	TjModel M;
	M.L = L;
	M.nup = nup;
	M.ndown = ndown;
	const size_t LL = (size_t)L * L;
	M.hop_re.assign(hop_re, hop_re + LL);
	if (hop_im) {
		M.hop_im.assign(hop_im, hop_im + LL);
		for (size_t k = 0; k < LL; k++) M.has_im |= hop_im[k] != 0;
	}
	M.jpm.assign(jpm, jpm + LL);
	TjPlan P;
	bool ok = false;
	std::string why;
	tj_plan(M, P, &ok, &why);
	if (!ok) return LPP_OK;
	out[1] = P.nblk;
	out[2] = P.ns;
	out[3] = P.kbits;
	out[4] = (int64_t)P.items.size();
	out[5] = (int64_t)P.pairs.size();
	out[6] = (int64_t)P.hops.size();
	for (const TjBlock& B : P.blocks) {
		out[7] += B.nxl;
		out[8] = std::max<int64_t>(out[8], B.nx);
		out[9] = std::max<int64_t>(out[9], B.nh);
	}
	if (!rowptr) { // statistics only
		out[0] = 1;
		return LPP_OK;
	}
	if (!colind || !values || nrows != (int64_t)P.nblk * P.ns || (M.has_im && !is_complex)) return LPP_OK;
	const int Lo = P.Lo;
	const uint32_t lbm = (1u << P.lb) - 1u;
	auto rank_of = [&](uint32_t s) -> int { return P.hi_base[s >> P.lb] + (int)P.lo_rank[s & lbm]; };
	auto rank_comb = [&](uint32_t w) -> int64_t { // BasisOneSpin.h:73-81
		int64_t r = 0;
		int c = 1;
		while (w) {
			const int b = __builtin_ctz(w);
			r += (int64_t)binom_t(b, c++);
			w &= w - 1;
		}
		return r;
	};
	const int64_t cfree = (int64_t)binom_t(L - ndown, nup);
	// the reference's index of (hole set, spin pattern): down word major, the up word ranked among the sites the down word leaves free
	auto basis_index = [&](int b, uint32_t sg) -> int64_t {
		const uint32_t hm = P.holes[(size_t)b];
		uint32_t up = 0, dn = 0;
		int k = 0;
		for (int i = 0; i < L; i++) {
			if ((hm >> i) & 1u) continue;
			if ((sg >> k) & 1u) up |= 1u << i;
			else dn |= 1u << i;
			k++;
		}
		uint32_t upc = 0; // the up word compressed onto the sites without a down electron
		int q = 0;
		for (int i = 0; i < L; i++) {
			if ((dn >> i) & 1u) continue;
			if ((up >> i) & 1u) upc |= 1u << q;
			q++;
		}
		return rank_comb(upc) + rank_comb(dn) * cfree;
	};
	struct Ent {
		int64_t col;
		double re, im;
	};
	std::vector<Ent> row;
	const int ncomp = is_complex ? 2 : 1;
	const double* va = (const double*)values;
	for (int b = 0; b < P.nblk; b++) {
		const TjBlock& B = P.blocks[(size_t)b];
		for (int r = 0; r < P.ns; r++) {
			const uint32_t sg = P.pat[(size_t)r];
			if (rank_of(sg) != r) return LPP_OK; // the rank tables
			row.clear();
			for (int e = 0; e < B.nx; e++) {
				const TjPair& pr = P.pairs[(size_t)(B.x_first + e)];
				if (__builtin_popcount(sg & pr.mask) != 1) continue;
				if (e < B.nxl && (pr.mask >> P.kbits)) return LPP_OK; // a "low" bond that is not
				row.push_back(Ent { basis_index(b, sg ^ pr.mask), pr.v, 0.0 });
			}
			for (int h = 0; h < B.nh; h++) {
				const TjHop& hp = P.hops[(size_t)(B.h_first + h)];
				const int lo = hp.lo, m = hp.m;
				const uint32_t wm = (2u << m) - 1u, seg = (sg >> lo) & wm;
				uint32_t bit, nseg, ups;
				if (hp.dir == 0) {
					bit = seg & 1u;
					nseg = (seg >> 1) | (bit << m);
					ups = (uint32_t)__builtin_popcount(seg >> 1);
				} else {
					bit = (seg >> m) & 1u;
					nseg = ((seg << 1) & wm) | bit;
					ups = (uint32_t)__builtin_popcount(seg & (wm >> 1));
				}
				const bool minus = ((bit ? ups : (uint32_t)m - ups) & 1u) != 0;
				const uint32_t s2 = (sg & ~(wm << lo)) | (nseg << lo);
				if (__builtin_popcount(s2) != nup || lo + m >= Lo) return LPP_OK;
				row.push_back(Ent { basis_index(hp.dst, s2), minus ? -hp.vr : hp.vr, minus ? -hp.vi : hp.vi });
			}
			std::sort(row.begin(), row.end(), [](const Ent& x, const Ent& y) { return x.col < y.col; });
			const int64_t ri = basis_index(b, sg);
			if (ri < 0 || ri >= nrows) return LPP_OK;
			size_t k = 0;
			for (int64_t p = rowptr[ri]; p < rowptr[ri + 1]; p++) {
				if (colind[p] == ri) continue; // the diagonal comes from the device assembler (diag_of), not from the plan
				if (k >= row.size() || row[k].col != colind[p]) return LPP_OK;
				const double im = is_complex ? va[p * ncomp + 1] : 0.0;
				if (std::memcmp(&row[k].re, &va[p * ncomp], 8) != 0 || (is_complex && std::memcmp(&row[k].im, &im, 8) != 0 && !(row[k].im == 0.0 && im == 0.0))) return LPP_OK;
				k++;
			}
			if (k != row.size()) return LPP_OK;
		}
	}
	out[0] = 1;
	return LPP_OK;
}
