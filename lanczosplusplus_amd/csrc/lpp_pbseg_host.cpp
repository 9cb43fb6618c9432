// lpp_pbseg_host.cpp -- host part of the segmented in-block form (lpp_pbseg.h): reads the species' basis (L sites, n particles) and
// the hopping amplitudes off the in-block matrix T it is given, cuts the row into segments by the top s sites, and packs the three
// kinds of hops.  Nothing is assumed about the lattice: T is decoded entry by entry, and the packed description is expanded again
// and compared with T bit by bit before it is used (seg_verify); anything that does not fit returns *ok = false and the caller keeps
// the per-position template of lpp_pbig_kernels.h.
#include <algorithm>
#include <cstring>
#include <map>

#include "lpp_host.h"
#include "lpp_pbseg.h"

namespace lpp {

namespace {

struct Binom {
	uint64_t c[34][34];
	Binom()
	{
		for (int a = 0; a < 34; a++)
			for (int b = 0; b < 34; b++) c[a][b] = b == 0 ? 1 : (a == 0 ? 0 : std::min<uint64_t>(c[a - 1][b - 1] + c[a - 1][b], (uint64_t)1 << 40));
	}
};
const Binom kBinom;

// rank of a word among the words of the same popcount in ascending order (BasisOneSpin.h:73-81)
inline int64_t rank_of(uint32_t w)
{
	int64_t r = 0;
	int k = 0;
	while (w) {
		const int b = __builtin_ctz(w);
		k++;
		r += (int64_t)kBinom.c[b][k];
		w &= w - 1;
	}
	return r;
}

void words_of(int L, int n, std::vector<uint32_t>& out)
{
	out.clear();
	if (n == 0) {
		out.push_back(0);
		return;
	}
	uint64_t v = ((uint64_t)1 << n) - 1;
	const uint64_t lim = (uint64_t)1 << L;
	while (v < lim) {
		out.push_back((uint32_t)v);
		const uint64_t c = v & (~v + 1), r = v + c;
		v = (((r ^ v) >> 2) / c) | r;
	}
}

inline int parity(uint32_t x) { return __builtin_popcount(x) & 1; }
inline uint32_t between(int a, int b) // bits strictly between sites a and b
{
	const int lo = std::min(a, b), hi = std::max(a, b);
	return (uint32_t)((((uint64_t)1 << hi) - 1) & ~(((uint64_t)1 << (lo + 1)) - 1));
}
inline bool same_bits(double a, double b) { return std::memcmp(&a, &b, 8) == 0; }

} // namespace

namespace {

// steps 1-2 of the plan: the basis (L sites, n particles: the candidate whose ascending words explain every entry of T as one hop) and the
// hopping amplitudes hv[to][from] -- every entry (r, c) is hv * (-1)^(particles strictly between the two sites) -- with the number of
// entries that carry each.  false: T is not the hopping matrix of one species in that basis.
bool seg_decode(int64_t n_up, const int64_t* rp, const int32_t* ci, const double* va, int& L, int& n, std::vector<double>& hv, std::vector<int64_t>& cnt)
{
	std::vector<uint32_t> W;
	L = n = 0;
	for (int l = 2; l <= 30 && L == 0; l++)
		for (int k = 1; k < l && L == 0; k++) {
			if ((int64_t)kBinom.c[l][k] != n_up) continue;
			words_of(l, k, W);
			bool good = true;
			const int64_t step = std::max<int64_t>(1, n_up / 257); // a sample decides between C(L, n) and C(L, L - n); every entry is checked below
			for (int64_t r = 0; r < n_up && good; r += step)
				for (int64_t p = rp[r]; p < rp[r + 1] && good; p++) {
					if (ci[p] == r) continue;
					if (ci[p] < 0 || ci[p] >= n_up) return false;
					const uint32_t x = W[(size_t)r] ^ W[(size_t)ci[p]];
					good = __builtin_popcount(x) == 2 && __builtin_popcount(x & W[(size_t)r]) == 1;
				}
			if (good) {
				L = l;
				n = k;
			}
		}
	if (L == 0) return false;
	hv.assign((size_t)L * L, 0.0);
	cnt.assign((size_t)L * L, 0);
	for (int64_t r = 0; r < n_up; r++)
		for (int64_t p = rp[r]; p < rp[r + 1]; p++) {
			const int64_t c = ci[p];
			if (c == r) continue; // the diagonal lives in D
			if (c < 0 || c >= n_up) return false;
			const uint32_t wr = W[(size_t)r], wc = W[(size_t)c], x = wr ^ wc;
			if (__builtin_popcount(x) != 2 || __builtin_popcount(x & wr) != 1) return false;
			const int to = __builtin_ctz(x & wr), from = __builtin_ctz(x & wc);
			const double v = parity(wc & between(to, from)) ? -va[p] : va[p];
			if (v == 0.0 || v != v) return false;
			const size_t k = (size_t)to * L + from;
			if (cnt[k] == 0)
				hv[k] = v;
			else if (!same_bits(hv[k], v))
				return false;
			cnt[k]++;
		}
	const int64_t per_bond = n >= 1 && L >= 2 ? (int64_t)kBinom.c[L - 2][n - 1] : 0;
	for (size_t k = 0; k < cnt.size(); k++)
		if (cnt[k] != 0 && cnt[k] != per_bond) return false; // a hop that is allowed must be there (CSR rows hold a column once)
	return true;
}

} // namespace

// steps 3-7: the plan from (L, n, amplitudes); cnt[to * L + from] != 0 marks the hops that exist
lpp_status pb_seg_plan_model(int L, int n, const std::vector<double>& hv, const std::vector<int64_t>& cnt, int wcap, SegPlan& P, bool* ok, bool one_block)
{
	*ok = false;
	P = SegPlan();
	// (one block per workgroup -- chains -- forms 64-bit addresses: rows up to 2^31 positions, 32 sites, 17 high sites)
	if (L < 2 || L > (one_block ? 32 : 30) || n < 1 || n >= L || wcap < 64 || wcap > 8128) return LPP_OK;
	const int64_t n_up = (int64_t)kBinom.c[L][n];
	if (n_up < 128 || n_up > (one_block ? (int64_t)2000000000 : (int64_t)400000000)) return LPP_OK; // 32-bit byte offsets into a row
	// ---- 3. the cut: smallest s whose longest segment fits the window --------------------------------------------------------
	int s = 0;
	int64_t maxseg = 0;
	for (int t = 1; t < L && s == 0; t++) {
		int64_t m = 0;
		for (int p = 0; p <= std::min(t, n); p++)
			if (n - p <= L - t) m = std::max<int64_t>(m, (int64_t)kBinom.c[L - t][n - p]);
		if (m <= wcap) {
			s = t;
			maxseg = m;
		}
	}
	if (s == 0 || s > (one_block ? 17 : 16) || L - s < 1) return LPP_OK;
	const int Ll = L - s; // low sites 0..Ll-1, high sites Ll..L-1
	P.L = L;
	P.n = n;
	P.s = s;
	P.n_up = n_up;
	// ---- 4. segments, stored order (by length, then by t), permutation ------------------------------------------------------------
	struct Seg {
		int t, k; // high configuration, low particles
		int64_t len, nbase; // natural position of its first element
	};
	std::vector<Seg> segs;
	for (int t = 0; t < (1 << s); t++) {
		const int k = n - __builtin_popcount((unsigned)t);
		if (k < 0 || k > Ll) continue;
		const uint32_t first = ((uint32_t)t << Ll) | (k ? (uint32_t)(((uint64_t)1 << k) - 1) : 0u);
		segs.push_back(Seg { t, k, (int64_t)kBinom.c[Ll][k], rank_of(first) });
	}
	std::stable_sort(segs.begin(), segs.end(), [](const Seg& a, const Seg& b) { return a.len > b.len; });
	std::vector<int> seg_of_t((size_t)1 << s, -1);
	std::vector<int64_t> sbase(segs.size() + 1, 0);
	for (size_t i = 0; i < segs.size(); i++) {
		seg_of_t[(size_t)segs[i].t] = (int)i;
		sbase[i + 1] = sbase[i] + segs[i].len;
	}
	if (sbase[segs.size()] != n_up) return LPP_OK;
	P.perm.resize((size_t)n_up);
	P.inv.resize((size_t)n_up);
	for (size_t i = 0; i < segs.size(); i++)
		for (int64_t o = 0; o < segs[i].len; o++) {
			P.perm[(size_t)(sbase[i] + o)] = (int32_t)(segs[i].nbase + o);
			P.inv[(size_t)(segs[i].nbase + o)] = (int32_t)(sbase[i] + o);
		}
	// low words per class
	std::vector<std::vector<uint32_t>> lows((size_t)Ll + 1);
	for (int k = 0; k <= Ll; k++)
		if (k <= n && n - k <= s) words_of(Ll, k, lows[(size_t)k]);
	// value groups of the low-low hops
	std::vector<uint64_t> keys;
	for (int to = 0; to < Ll; to++)
		for (int from = 0; from < Ll; from++) {
			if (cnt[(size_t)to * L + from] == 0) continue;
			for (int sg = 0; sg < 2; sg++) {
				const double v = sg ? -hv[(size_t)to * L + from] : hv[(size_t)to * L + from];
				uint64_t key;
				std::memcpy(&key, &v, 8);
				if (std::find(keys.begin(), keys.end(), key) == keys.end()) keys.push_back(key);
			}
		}
	std::sort(keys.begin(), keys.end());
	// (a species without low-low hops keeps one group that no entry uses)
	if (keys.size() > 2) return LPP_OK; // the kernel unrolls one or two value groups (+-t); anything else keeps the per-position template
	P.G = std::max<int>(1, (int)keys.size());
	for (size_t g = 0; g < keys.size(); g++) std::memcpy(&P.gval[g], &keys[g], 8);
	const int G = P.G;
	// ---- 5. cross tables (class, low site, add / remove), built on demand ----------------------------------------------------------
	std::map<int, int32_t> table_at; // key = ((k * 33 + site a) * 33 + site b + 1) * 2 + dir
	auto table = [&](int k, int sa, int sb, int dir) -> int32_t { // dir 0: the source has one MORE low particle (at the site), 1: one less; sb = -1: one hop only
		const int key = ((k * 33 + sa) * 33 + sb + 1) * 2 + dir;
		auto it = table_at.find(key);
		if (it != table_at.end()) return it->second;
		const std::vector<uint32_t>& lw = lows[(size_t)k];
		const size_t len64 = (lw.size() + 63) & ~(size_t)63;
		const int32_t at = (int32_t)P.xwords.size();
		P.xwords.resize(P.xwords.size() + len64, 0);
		for (int half = 0; half < 2; half++) {
			const int site = half ? sb : sa;
			if (site < 0) continue;
			for (size_t o = 0; o < lw.size(); o++) {
				const uint32_t w = lw[o];
				const bool has = (w >> site) & 1u;
				if (dir == 0 ? has : !has) continue;
				const uint32_t src = dir == 0 ? (w | (1u << site)) : (w & ~(1u << site));
				const int64_t off = rank_of(src);
				const bool minus = parity(w >> (site + 1)); // low particles above the site (between it and the high site)
				P.xwords[(size_t)at + o] |= (uint32_t)((uint16_t)off | (minus ? kSegWordMinus : kSegWordPlus)) << (16 * half);
			}
			// positions without this hop still load an element (value 0): the one a neighbouring lane of their slice loads anyway, so that
			// they ask for no cache line of their own (the kernel is bound by the lines a slice requests from L1)
			for (size_t j0 = 0; j0 < len64; j0 += 64) {
				uint32_t fill = 0;
				bool any = false;
				for (size_t o = j0; o < j0 + 64 && !any; o++)
					if ((P.xwords[(size_t)at + o] >> (16 * half)) & 0x4000u) {
						fill = (P.xwords[(size_t)at + o] >> (16 * half)) & 0x1fffu;
						any = true;
					}
				for (size_t o = j0; o < j0 + 64; o++) {
					const uint32_t w = (P.xwords[(size_t)at + o] >> (16 * half)) & 0xffffu;
					if (w & 0x4000u)
						fill = w & 0x1fffu;
					else
						P.xwords[(size_t)at + o] |= fill << (16 * half);
				}
			}
		}
		table_at[key] = at;
		return at;
	};
	// ---- 6. per-segment scalars ----------------------------------------------------------------------------------------------------
	P.segs.resize(segs.size());
	for (size_t i = 0; i < segs.size(); i++) {
		const int t = segs[i].t, k = segs[i].k;
		SegInst& S = P.segs[i];
		S = SegInst();
		S.sbase = (int32_t)sbase[i];
		S.len = (int32_t)segs[i].len;
		S.cross_first = (int32_t)P.cross.size();
		for (int jh = 0; jh < s; jh++) {
			const int j = Ll + jh;
			const bool has = (t >> jh) & 1;
			const int t2 = t ^ (1 << jh);
			const int k2 = has ? k + 1 : k - 1; // low particles of the source segment
			if (k2 < 0 || k2 > Ll || seg_of_t[(size_t)t2] < 0) continue;
			const double sh = parity((uint32_t)t & ((1u << jh) - 1)) ? -1.0 : 1.0; // high particles below the high site
			int prev = -1;
			double prev_v = 0.0;
			for (int i2 = 0; i2 <= Ll; i2++) {
				// row has the particle at the high site j: it came from low site i2 (hv[j][i2]); otherwise the row has it at i2 and it came from j
				const size_t kk = i2 < Ll ? (has ? (size_t)j * L + i2 : (size_t)i2 * L + j) : 0;
				if (i2 < Ll && cnt[kk] == 0) continue;
				if (i2 < Ll && prev < 0) { // first of a pair
					prev = i2;
					prev_v = sh * hv[kk];
					continue;
				}
				if (prev < 0) break; // i2 == Ll, nothing pending
				SegCross c;
				c.wordoff = table(k, prev, i2 < Ll ? i2 : -1, has ? 0 : 1);
				c.srcbase = (int32_t)sbase[(size_t)seg_of_t[(size_t)t2]];
				c.val[0] = prev_v;
				c.val[1] = i2 < Ll ? sh * hv[kk] : 0.0;
				c.pad = 0;
				P.cross.push_back(c);
				prev = -1;
			}
		}
		S.ncross = (int32_t)P.cross.size() - S.cross_first;
		S.hh_first = (int32_t)P.hh.size();
		for (int a = 0; a < s; a++)
			for (int b = 0; b < s; b++) {
				if (a == b || !((t >> a) & 1) || ((t >> b) & 1)) continue; // row has a, lacks b: the source had it at b
				const size_t kk = (size_t)(Ll + a) * L + (Ll + b);
				if (cnt[kk] == 0) continue;
				const int t2 = t ^ (1 << a) ^ (1 << b);
				if (seg_of_t[(size_t)t2] < 0) continue;
				SegHh h;
				h.srcbase = (int32_t)sbase[(size_t)seg_of_t[(size_t)t2]];
				h.pad = 1; // lane l reads element srcbase + offset in the segment
				// the kernel reads a lone position as the SECOND element of the pair in front of it (k_pb_up_seg, issue_data): a one-position source
				// segment at the very start of the row has no element in front of it -- such a space keeps the per-position template
				if (h.srcbase == 0 && segs[i].len == 1) return LPP_OK;
				h.val = parity((uint32_t)t & (between(a, b))) ? -hv[kk] : hv[kk];
				P.hh.push_back(h);
			}
		S.nhh = (int32_t)P.hh.size() - S.hh_first;
		if (S.ncross > kSegMaxCross || S.nhh > (one_block ? kSegMaxHh : 12)) return LPP_OK;
		P.max_cross = std::max(P.max_cross, (int)S.ncross);
		P.max_hh = std::max(P.max_hh, (int)S.nhh);
	}
	// ---- 7. items (consecutive segments sharing a window) and their types ------------------------------------------------------
	const int64_t cap = maxseg >= 4096 ? maxseg : wcap;
	struct TypeInfo {
		int32_t slice_first, nslices, zero_at;
	};
	std::map<std::vector<int32_t>, int> type_of; // key: the low-particle numbers of the item's segments
	std::vector<TypeInfo> types;
	std::vector<std::pair<int, int>> edges;
	std::vector<int> ecol, colour, slot_idx;
	for (size_t i0 = 0; i0 < segs.size();) {
		size_t i1 = i0;
		int64_t wl = 0;
		while (i1 < segs.size() && (int)(i1 - i0) < kSegMaxSegs && (i1 - i0) < 255 && (i1 == i0 || wl + segs[i1].len <= cap)) wl += segs[i1++].len;
		std::vector<int32_t> key;
		for (size_t i = i0; i < i1; i++) key.push_back(segs[i].k);
		auto it = type_of.find(key);
		int ty;
		if (it != type_of.end())
			ty = it->second;
		else {
			ty = (int)types.size();
			type_of[key] = ty;
			TypeInfo TI;
			TI.slice_first = (int32_t)P.slices.size();
			TI.zero_at = (int32_t)((kSegWinPad + wl + 2 + 31) & ~(int64_t)31);
			if (TI.zero_at + kPbZeroSlotsHost > 8192) return LPP_OK; // 16-bit window indices x 8 stay below 64 KB
			// slices: 64 consecutive positions of one segment; in-window lists per (slice, group)
			int64_t oa = 0;
			for (size_t i = i0; i < i1; i++) {
				const int k = segs[i].k;
				const std::vector<uint32_t>& lw = lows[(size_t)k];
				for (int64_t j0 = 0; j0 < segs[i].len; j0 += 64) {
					SegSlice sl;
					sl.first = (uint16_t)(oa + j0);
					sl.count = (uint8_t)std::min<int64_t>(64, segs[i].len - j0);
					sl.seg = (uint8_t)(i - i0);
					sl.segoff = (uint16_t)j0;
					sl.nc[0] = sl.nc[1] = 0;
					sl.off[0] = sl.off[1] = 0;
					for (int g = 0; g < G; g++) {
						int nslots = 0;
						struct Ent {
							int lane, col, slot;
						};
						std::vector<Ent> ents;
						for (int h = 0; h < 2; h++) {
							edges.clear();
							ecol.clear();
							int seen[32] = { 0 };
							for (int l = 0; l < 32; l++) {
								const int64_t o = j0 + h * 32 + l;
								if (o >= segs[i].len) break;
								const uint32_t w = lw[(size_t)o];
								for (int to = 0; to < Ll; to++) {
									if (!((w >> to) & 1u)) continue;
									for (int from = 0; from < Ll; from++) {
										if (((w >> from) & 1u) || cnt[(size_t)to * L + from] == 0) continue;
										const double v = parity(w & between(to, from)) ? -hv[(size_t)to * L + from] : hv[(size_t)to * L + from];
										uint64_t kv;
										std::memcpy(&kv, &v, 8);
										if (kv != keys[(size_t)g]) continue;
										const uint32_t src = (w & ~(1u << to)) | (1u << from);
										const int col = (int)(kSegWinPad + oa + rank_of(src));
										const int bank = col & 31;
										edges.emplace_back(l, bank * 2 + (seen[bank]++ % 2)); // a bank serves two addresses per slot (pb_pack_template, bank_ways = 2)
										ecol.push_back(col);
									}
								}
							}
							const int D = edge_colour(edges, 64, colour);
							nslots = std::max(nslots, D);
							for (size_t e = 0; e < edges.size(); e++) ents.push_back(Ent { h * 32 + edges[e].first, ecol[e], colour[e] });
							P.entries_lo += (int64_t)edges.size();
						}
						const int nchunks = (nslots + 3) / 4;
						nslots = nchunks * 4;
						slot_idx.assign((size_t)nslots * 64, -1);
						for (const Ent& e : ents) slot_idx[(size_t)e.slot * 64 + e.lane] = e.col;
						for (int sl2 = 0; sl2 < nslots; sl2++)
							for (int h = 0; h < 2; h++) { // filling: the zero slot in the bank this slot and half-wave use least
								int used[32] = { 0 };
								bool any_pad = false;
								for (int l = 0; l < 32; l++) {
									const int c = slot_idx[(size_t)sl2 * 64 + h * 32 + l];
									if (c >= 0)
										used[c & 31]++;
									else
										any_pad = true;
								}
								if (!any_pad) continue;
								int z = 0;
								for (int t2 = 1; t2 < 32; t2++)
									if (used[(TI.zero_at + t2) & 31] < used[(TI.zero_at + z) & 31]) z = t2;
								for (int l = 0; l < 32; l++)
									if (slot_idx[(size_t)sl2 * 64 + h * 32 + l] < 0) slot_idx[(size_t)sl2 * 64 + h * 32 + l] = TI.zero_at + z;
							}
						if (nchunks > 254) return LPP_OK;
						// two chunks per 16-byte load: pair cp holds, per lane, the two words of chunk 2 cp and the two of chunk 2 cp + 1 (filling: zero slots)
						sl.off[g] = (int32_t)(P.words.size() / 256);
						sl.nc[g] = (uint8_t)nchunks;
						for (int c = 0; c < ((nchunks + 1) & ~1); c += 2)
							for (int l = 0; l < 64; l++)
								for (int q = 0; q < 8; q += 2) {
									const int sl0 = 4 * c + q;
									const uint32_t lo16 = sl0 < nslots ? (uint32_t)slot_idx[(size_t)sl0 * 64 + l] : (uint32_t)TI.zero_at;
									const uint32_t hi16 = sl0 + 1 < nslots ? (uint32_t)slot_idx[(size_t)(sl0 + 1) * 64 + l] : (uint32_t)TI.zero_at;
									P.words.push_back(lo16 | (hi16 << 16));
								}
						P.slots_lo += (int64_t)nslots * 64;
					}
					P.slices.push_back(sl);
				}
				oa += segs[i].len;
			}
			TI.nslices = (int32_t)P.slices.size() - TI.slice_first;
			if (TI.nslices > kSegMaxSlices) return LPP_OK;
			types.push_back(TI);
		}
		SegItem I;
		I.c0 = (int32_t)sbase[i0];
		I.wlen = (int32_t)wl;
		I.seg_first = (int32_t)i0;
		I.nseg = (int32_t)(i1 - i0);
		I.slice_first = types[(size_t)ty].slice_first;
		I.nslices = types[(size_t)ty].nslices;
		I.zero_at = types[(size_t)ty].zero_at;
		I.type = ty;
		P.items.push_back(I);
		P.wmax = std::max<int>(P.wmax, (int)wl);
		P.zmax = std::max<int>(P.zmax, I.zero_at);
		i0 = i1;
	}
	P.ntypes = (int)types.size();
	if (G == 2) { // a value group no low-low entry uses (an open chain: nothing ever sits between neighbours, only +hv occurs) is dropped
		int64_t used[2] = { 0, 0 };
		for (const SegSlice& sl : P.slices) {
			used[0] += sl.nc[0];
			used[1] += sl.nc[1];
		}
		if (used[0] == 0 || used[1] == 0) {
			const int keep = used[0] == 0 ? 1 : 0;
			for (SegSlice& sl : P.slices) {
				sl.nc[0] = sl.nc[keep];
				sl.off[0] = sl.off[keep];
				sl.nc[1] = 0;
				sl.off[1] = 0;
			}
			P.gval[0] = P.gval[keep];
			P.gval[1] = 0.0;
			P.G = 1;
		}
	}
	if (P.G == 2) { // group 0 requested two chunks ahead when its lists rarely hold more (4x5 lattice: the + hops hold 2, the - hops 3-4 chunks)
		int64_t longer = 0;
		for (size_t j = 0; j < P.slices.size(); j++) longer += P.slices[j].nc[0] > 2 ? 1 : 0;
		if (longer * 8 <= (int64_t)P.slices.size()) P.pre0 = 2;
	}
	P.ws = P.zmax + kPbZeroSlotsHost; // window stride in elements (even)
	P.words.resize(P.words.size() + 256 * 4, (uint32_t)P.zmax | ((uint32_t)P.zmax << 16)); // slack for the look-ahead loads
	const int32_t zero_table = (int32_t)P.xwords.size(); // 8192 + 64 zero words: the table of the filling entries (any offset of any segment stays inside)
	P.xwords.resize(P.xwords.size() + 8192 + 64, 0);
	// every segment's lists padded to the kernel instance's width with entries of value 0.0 that read valid addresses: the slice
	// loop then carries no condition at all (padded cross entries: a table of the segment's own class -- or table 0, the longest --
	// and the row's first elements; padded high-high entries: the segment itself)
	// kernel instances (NC, NH): (2, 2), (5, 4), (6, 8) and -- one block per workgroup only: chains -- (1, 12), (2, 12)
	if (one_block) { // k_pb_up_seg<..., ROWS = 1> reads the first hop of a pair only: a chain's high sites have one low neighbour each
		for (const SegCross& c : P.cross)
			if (c.val[1] != 0.0) return LPP_OK;
	}
	if (P.max_cross <= 2 && P.max_hh <= 2) P.nc_pad = 2, P.nh_pad = 2;
	else if (one_block && P.max_cross <= 1) P.nc_pad = 1, P.nh_pad = P.max_hh <= 12 ? 12 : 16; // an open chain: the one bond between the low and the high sites
	else if (one_block && P.max_cross <= 2) P.nc_pad = 2, P.nh_pad = P.max_hh <= 12 ? 12 : 16; // ... and the bond between the two ends
	else if (one_block) return LPP_OK;
	else if (P.max_cross <= 5 && P.max_hh <= 4) P.nc_pad = 5, P.nh_pad = 4;
	else if (P.max_cross <= 6 && P.max_hh <= 8) P.nc_pad = 6, P.nh_pad = 8;
	else return LPP_OK;
	{
		std::vector<SegCross> cp(P.segs.size() * (size_t)P.nc_pad);
		std::vector<SegHh> hp(P.segs.size() * (size_t)P.nh_pad);
		for (size_t i = 0; i < P.segs.size(); i++) {
			SegInst& S = P.segs[i];
			for (int b = 0; b < P.nc_pad; b++)
				cp[i * P.nc_pad + b] = b < S.ncross ? P.cross[(size_t)(S.cross_first + b)] : SegCross { zero_table, 0, { 0.0, 0.0 }, 0 }; // all-zero words: every lane reads the row's first element (one line)
			for (int b = 0; b < P.nh_pad; b++) hp[i * P.nh_pad + b] = b < S.nhh ? P.hh[(size_t)(S.hh_first + b)] : SegHh { std::max(S.sbase - 1, 0), 0, 0.0 }; // pad = 0: every lane reads the same pair of elements, the segment's first and the one in front of it (one line, inside the row)
			S.cross_first = (int32_t)(i * P.nc_pad);
			S.hh_first = (int32_t)(i * P.nh_pad);
		}
		P.cross.swap(cp);
		P.hh.swap(hp);
		// the high-high hops an item's slices carry: the longest list of its segments in whole fours (bits 16.. of the item's type; the
		// one-block kernel runs the instance's loop for 4, 8 or 12 of them -- on a chain of 28 sites 3.9 instead of 6 loads per slice)
		for (SegItem& I : P.items) {
			int m = 0;
			for (int k = 0; k < I.nseg; k++) m = std::max(m, (int)P.segs[(size_t)(I.seg_first + k)].nhh);
			I.type |= std::min(P.nh_pad, std::max(4, (m + 3) & ~3)) << 16;
		}
	}
	*ok = true;
	return LPP_OK;
}

namespace {
// step 8: expand the packed description again and compare it with T, entry by entry and bit by bit
bool seg_verify(SegPlan& P, const int64_t* rp, const int32_t* ci, const double* va)
{
	const int G = P.G;
	const int64_t n_up = P.n_up;
	{
		std::vector<std::pair<int32_t, double>> row;
		for (const SegItem& I : P.items)
			for (int sj = 0; sj < I.nslices; sj++) {
				const SegSlice& sl = P.slices[(size_t)(I.slice_first + sj)];
				const SegInst& S = P.segs[(size_t)(I.seg_first + sl.seg)];
				for (int l = 0; l < sl.count; l++) {
					row.clear();
					for (int g = 0; g < G; g++) {
						for (int c = 0; c < sl.nc[g]; c++)
							for (int q = 0; q < 4; q++) {
								const uint32_t w = P.words[((size_t)(sl.off[g] + (c >> 1)) * 64 + l) * 4 + (c & 1) * 2 + (q >> 1)];
								const int idx = (int)((q & 1) ? (w >> 16) : (w & 0xffffu));
								if (idx >= I.zero_at) continue; // filling
								const int64_t sp = (int64_t)I.c0 + idx - kSegWinPad;
								if (sp < I.c0 || sp >= (int64_t)I.c0 + I.wlen) return false;
								row.emplace_back(P.perm[(size_t)sp], P.gval[g]);
							}
					}
					for (int b = 0; b < P.nc_pad; b++) { // the padded lists, as the kernel walks them
						const SegCross& cx = P.cross[(size_t)(S.cross_first + b)];
						for (int half = 0; half < 2; half++) {
							if (cx.val[half] == 0.0) continue;
							const uint32_t w = P.xwords[(size_t)cx.wordoff + sl.segoff + l] >> (16 * half);
							if (!(w & 0x4000)) continue;
							row.emplace_back(P.perm[(size_t)(cx.srcbase + (w & 0x1fff))], (w & 0x8000) ? -cx.val[half] : cx.val[half]);
						}
					}
					for (int b = 0; b < P.nh_pad; b++) {
						const SegHh& h = P.hh[(size_t)(S.hh_first + b)];
						if (h.val == 0.0) continue;
						row.emplace_back(P.perm[(size_t)(h.srcbase + sl.segoff + l)], h.val);
					}
					std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
					const int64_t r = P.perm[(size_t)(S.sbase + sl.segoff + l)];
					size_t k = 0;
					for (int64_t p = rp[r]; p < rp[r + 1]; p++) {
						if (ci[p] == r) continue;
						if (k >= row.size() || row[k].first != ci[p] || !same_bits(row[k].second, va[p])) return false;
						k++;
					}
					if (k != row.size()) return false;
					P.entries++;
				}
			}
		if (P.entries != n_up) return false; // every row was visited once
	}
	return true;
}
} // namespace

lpp_status pb_seg_plan(int64_t n_up, const int64_t* rp, const int32_t* ci, const double* va, int wcap, SegPlan& P, bool* ok)
{
	*ok = false;
	P = SegPlan();
	if (n_up < 128 || n_up >= ((int64_t)1 << 24)) return LPP_OK; // (T is walked on the host: one-species spaces of a Hubbard model)
	int L = 0, n = 0;
	std::vector<double> hv;
	std::vector<int64_t> cnt;
	if (!seg_decode(n_up, rp, ci, va, L, n, hv, cnt)) return LPP_OK;
	bool built = false;
	const lpp_status st = pb_seg_plan_model(L, n, hv, cnt, wcap, P, &built);
	if (st != LPP_OK || !built) return st;
	*ok = seg_verify(P, rp, ci, va);
	return LPP_OK;
}

} // namespace lpp

extern "C" lpp_status lpp_pb_seg_plan_stats(int64_t rows, const int64_t* rowptr, const int32_t* colind, const double* values, int32_t wcap, int64_t* out, int32_t* perm)
{
	if (!rowptr || !colind || !values || !out) return lpp::fail(LPP_ERR_INVALID, "lpp_pb_seg_plan_stats: null argument");
	lpp::SegPlan P;
	bool ok = false;
	const lpp_status st = lpp::pb_seg_plan(rows, rowptr, colind, values, wcap, P, &ok);
	if (st != LPP_OK) return st;
	out[0] = ok ? 1 : 0;
	out[1] = P.L;
	out[2] = P.n;
	out[3] = P.s;
	out[4] = (int64_t)P.segs.size();
	out[5] = (int64_t)P.items.size();
	out[6] = P.ntypes;
	out[7] = P.wmax;
	out[8] = (int64_t)(P.words.size() * 4 + P.xwords.size() * 4 + P.slices.size() * 16);
	out[9] = (int64_t)(P.cross.size() * 32 + P.hh.size() * 16 + P.segs.size() * 32 + P.items.size() * 32);
	out[10] = P.entries_lo;
	out[11] = P.slots_lo;
	out[12] = P.G;
	out[13] = P.ws;
	if (ok && perm) std::memcpy(perm, P.perm.data(), sizeof(int32_t) * P.perm.size());
	return LPP_OK;
}
