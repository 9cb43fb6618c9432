// lpp_pb.hip -- host side of the product-basis stored layout (kernels and rationale: lpp_pb_kernels.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lpp_assemble_kernels.h"
#include "lpp_engine_impl.h"
#include "lpp_pbig_kernels.h"
#include "lpp_pbseg_kernels.h"

using namespace lpp;

namespace lpp {

int64_t pb_pitch_for(int64_t n_up) { return (n_up + 15) & ~(int64_t)15; }

void free_pb(lpp_engine* e)
{
	PbState& B = e->pb;
	for (void* p : { (void*)B.tw, (void*)B.tw_off, (void*)B.tw_len, (void*)B.t_ptr, (void*)B.t_col, (void*)B.t_val, (void*)B.c_ptr, (void*)B.c_col,
	                 (void*)B.c_code, (void*)B.order, (void*)B.down_image, (void*)B.pace, (void*)B.z, (void*)B.u, (void*)B.xy, (void*)B.dict, (void*)B.dcode, (void*)B.blockbase,
	                 (void*)B.fw, (void*)B.f_off, (void*)B.f_len, (void*)B.c_pstart, (void*)B.dval, (void*)B.perm, (void*)B.inv, (void*)B.cdict,
	                 B.seg_items, B.seg_segs, B.seg_cross, B.seg_hh, B.seg_slices, (void*)B.seg_tw, (void*)B.seg_xw })
		if (p) (void)hipFree(p);
	B = PbState();
	e->pitch = e->pitch_rows = e->pitch_blocks = 0;
}

namespace {
template <typename V> lpp_status to_device(V** dst, const std::vector<V>& src, hipStream_t st)
{
	HIP_TRY_MEM(hipMalloc(dst, sizeof(V) * std::max<size_t>(src.size(), 1)));
	if (!src.empty()) HIP_TRY(hipMemcpyAsync(*dst, src.data(), sizeof(V) * src.size(), hipMemcpyHostToDevice, st));
	return LPP_OK;
}

uint8_t code_of(const double* dict, int ndict, double v)
{
	// same order as the device's dict_code: bit patterns compared as unsigned integers
	uint64_t key;
	std::memcpy(&key, &v, 8);
	int lo = 0, hi = ndict - 1;
	while (lo < hi) {
		const int mid = (lo + hi) >> 1;
		uint64_t k;
		std::memcpy(&k, &dict[mid], 8);
		if (k < key)
			lo = mid + 1;
		else
			hi = mid;
	}
	return (uint8_t)lo;
}
} // namespace

// the realified in-block matrix: complex entry (i, c, t) -> row 2i: (2c, Re t), (2c+1, -Im t); row 2i+1: (2c, Im t), (2c+1, Re t); zero parts
// are dropped (a real hop costs two entries, not four); columns stay ascending
void pb_realify(int64_t n, const int64_t* rp, const int32_t* ci, const double* va, std::vector<int64_t>& rrp, std::vector<int32_t>& rci, std::vector<double>& rva)
{
	rrp.assign((size_t)(2 * n) + 1, 0);
	rci.clear();
	rva.clear();
	for (int64_t i = 0; i < n; i++)
		for (int half = 0; half < 2; half++) {
			for (int64_t p = rp[i]; p < rp[i + 1]; p++) {
				const int64_t c = ci[p];
				if (c == i) continue; // the diagonal lives in D
				const double re = va[2 * p], im = va[2 * p + 1];
				const double v0 = half == 0 ? re : im, v1 = half == 0 ? -im : re; // coefficients of (Re y_c, Im y_c)
				if (v0 != 0.0) {
					rci.push_back((int32_t)(2 * c));
					rva.push_back(v0);
				}
				if (v1 != 0.0) {
					rci.push_back((int32_t)(2 * c + 1));
					rva.push_back(v1);
				}
			}
			rrp[(size_t)(2 * i + half) + 1] = (int64_t)rci.size();
		}
}

lpp_status pb_build(lpp_engine* e, int64_t n_up, int64_t n_blk, const int64_t* t_rp, const int32_t* t_ci, const double* t_va,
                    const int64_t* c_rp, const int32_t* c_ci, const double* c_va, const double* dict256, int ndict,
                    int64_t blk0, int64_t nblk_loc, int64_t pitch_dn, int64_t nblk_padded, const PbCplxInput* cx, SegPlan* pre, int64_t pre_nnz)
{
	free_pb(e);
	PbState& B = e->pb;
	hipStream_t st = e->stream;
	const int64_t pitch = pb_pitch_for(n_up);
	const bool tx = pitch_dn > 0; // several GPUs: own blocks for the in-block part, the transposed slice for the couplings
	if (nblk_loc < 0) nblk_loc = n_blk;
	if (!tx) {
		pitch_dn = pitch;
		nblk_padded = n_blk;
	}
	if (blk0 < 0 || blk0 + nblk_loc > n_blk || (pitch_dn & 15) || nblk_padded < n_blk) return fail(LPP_ERR_INVALID, "pb_build: bad block range / coupling pitch");
	// Rows beyond one LDS window are cut into pieces (k_pb_up_big): two 512-thread workgroups of <= 80 KB per CU.  LPP_PB_PIECE_ROWS
	// forces pieces of (at most) that many positions on any matrix (tests run the small cases of the suite through them).
	int64_t wmax = 0;
	// pieces of <= 8128 positions: two blocks' windows share a workgroup (k_pb_up_big2) and the second one stays within the 64 KB an
	// LDS instruction's offset reaches; LPP_PB_BIG2=0: one block per workgroup, two 512-thread workgroups per CU (k_pb_up_big, <= 8768)
	const bool big2 = !(getenv("LPP_PB_BIG2") && atoi(getenv("LPP_PB_BIG2")) == 0);
	// round 5: four value groups (complex hoppings realified) two blocks at a time too -- half the template words per output row: 4.74 against
	// 5.48 ms (k_pb_up_big<.., 4, 3>) at 1.47e8 complex states; LPP_PB_BIG2_FOUR=0: the one-block kernel
	const bool big2_four = !(getenv("LPP_PB_BIG2_FOUR") && atoi(getenv("LPP_PB_BIG2_FOUR")) == 0);
	// one window needs the row, its diagonal codes and the template's list heads in LDS (pb_up_lds_bytes; two value groups assumed here):
	// rows of 17,400-19,500 positions pass a test of the row alone and then failed pb_build -- the 3x6 lattice's 6-electron species (18,564)
	if ((size_t)(pitch + kPbZeroSlots) * sizeof(double) > (size_t)156 * 1024 || pb_up_lds_bytes(pitch, (int)((n_up + 63) / 64), 2) > (size_t)160 * 1024 - 64)
		wmax = big2 ? 8128 : 8768;
	if (const char* s = getenv("LPP_PB_PIECE_ROWS")) wmax = std::max<int64_t>(64, std::min<int64_t>(atoll(s), big2 ? 8128 : 16384)) & ~(int64_t)63;
	int64_t W = 0;
	if (wmax > 0) {
		const int64_t np = (n_up + wmax - 1) / wmax;
		W = (((n_up + np - 1) / np) + 63) & ~(int64_t)63;
		if (n_up >= ((int64_t)1 << 24) && !pre) return fail(LPP_ERR_INVALID, "pb_build: more than 2^24 positions per block"); // (24-bit positions of the per-position template)
	}
	// Couplings: the panel of 16 positions of all blocks has to stay in one XCD's L2 (4 MiB) while it is gathered from.  Beyond
	// ~1.6 MB the source blocks are walked in parts (k_pb_down_parts, which also forms 64-bit addresses: vectors beyond 4 GiB).
	const size_t vec_bytes = (size_t)nblk_padded * (size_t)pitch_dn * sizeof(double);
	// (the pacing of k_pb_down_parts keeps the workgroups of a group within two consecutive parts: two parts of <= 1.7 MB live)
	// Measured at N_dn = 38760 (panel 4.96 MB, 3.0e9 states, scripts/experiments/r03_parts_ab.sh): whole panel 46.6 ms per product
	// (the misses of an over-full L2 are Infinity-Cache hits), 2 parts 51.4, 3 parts 56.7, 4 parts 63.4, 6 parts 84.3 -- the part
	// phases cost more (a wait per part, lists padded per part) than the L2 hits return.  So parts are kept for panels beyond
	// 6 MB only; up to there the whole-panel kernel runs, with 64-bit addresses where the vector needs them (k_pb_down<WIDE>).
	int nparts = (size_t)nblk_padded * 128 <= (size_t)6 << 20 || n_blk > 65535 ? 1 : 2; // (the parts form keeps 16-bit places and all of a workgroup's sums in registers: below 65536 blocks)
	if (const char* s = getenv("LPP_PB_PARTS")) nparts = std::max(1, std::min(atoi(s), kPbMaxParts));
	const bool parts = nparts > 1 || getenv("LPP_PB_PARTS") != nullptr;
	const bool wide = vec_bytes >= ((size_t)1 << 32) || getenv("LPP_PB_WIDE") != nullptr;
	PbTemplate T;
	int ways = 2;
	if (const char* s = getenv("LPP_PB_BANK_WAYS")) ways = std::max(1, std::min(atoi(s), 4));
	// Row order inside a block.  A slice of 64 rows walks as many template chunks per value group as its LONGEST list needs, and in
	// the basis order neighbouring rows have lists of very different lengths: at config 2 17.1 entries per row occupy 28.9 slots.
	// Stored in the order of their chunk counts (stable, so neighbours stay neighbours among equals) they occupy 20.2: 30 % fewer
	// template words through L1, LDS gathers and adds.  One-window, single-GPU form only: the pieces form lives on the locality of
	// the basis order and the exchange kernels address positions by their basis index.  LPP_PB_PERM=0 switches it off.
	std::vector<int32_t> perm, inv;
	std::vector<int64_t> p_rp;
	std::vector<int32_t> p_ci;
	std::vector<double> p_va;
	// complex hoppings: the realified in-block matrix is an ordinary real template (4-8 value groups: k_pb_up / k_pb_up_big with any number
	// of groups), in one window or in pieces; the couplings' kernel multiplies complex values (k_pb_down<CPLX>, with 64-bit addresses too)
	if (cx && (tx || parts)) return fail(LPP_ERR_INVALID, "pb_build: complex hoppings are held in the single-GPU forms with whole-panel couplings only");
	const bool want_perm = !cx && W == 0 && !tx && nblk_loc == n_blk && n_up >= 128 && !(getenv("LPP_PB_PERM") && atoi(getenv("LPP_PB_PERM")) == 0);
	if (want_perm) {
		std::vector<unsigned long long> vals; // distinct values, ascending bit pattern (the order pb_pack_template numbers its groups in does not matter here)
		for (int64_t i = 0; i < n_up && vals.size() <= (size_t)kPbGroupsMax; i++)
			for (int64_t q = t_rp[i]; q < t_rp[i + 1] && vals.size() <= (size_t)kPbGroupsMax; q++) {
				if (t_ci[q] == i) continue; // the diagonal is not part of the template (it lives in D)
				unsigned long long k;
				std::memcpy(&k, &t_va[q], 8);
				if (std::find(vals.begin(), vals.end(), k) == vals.end()) vals.push_back(k);
			}
		if (vals.size() <= (size_t)kPbGroupsMax && !vals.empty()) {
			const int G = (int)vals.size();
			std::vector<uint8_t> key((size_t)n_up * (size_t)G, 0); // chunks of 4 per group
			for (int64_t i = 0; i < n_up; i++) {
				int cnt[kPbGroupsMax] = { 0 };
				for (int64_t q = t_rp[i]; q < t_rp[i + 1]; q++) {
					if (t_ci[q] == i) continue;
					unsigned long long k;
					std::memcpy(&k, &t_va[q], 8);
					cnt[std::find(vals.begin(), vals.end(), k) - vals.begin()]++;
				}
				for (int g = 0; g < G; g++) key[(size_t)i * G + g] = (uint8_t)std::min(255, (cnt[g] + 3) / 4);
			}
			perm.resize((size_t)n_up);
			for (int64_t i = 0; i < n_up; i++) perm[(size_t)i] = (int32_t)i;
			std::stable_sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) { return std::lexicographical_compare(&key[(size_t)x * G], &key[(size_t)x * G] + G, &key[(size_t)y * G], &key[(size_t)y * G] + G); });
			inv.resize((size_t)n_up);
			for (int64_t q = 0; q < n_up; q++) inv[(size_t)perm[(size_t)q]] = (int32_t)q;
			// T' = P T P^T, columns ascending inside a row
			p_rp.assign((size_t)n_up + 1, 0);
			p_ci.resize((size_t)t_rp[n_up]);
			p_va.resize((size_t)t_rp[n_up]);
			std::vector<std::pair<int32_t, double>> row;
			for (int64_t q = 0; q < n_up; q++) {
				const int64_t i = perm[(size_t)q];
				row.clear();
				for (int64_t k = t_rp[i]; k < t_rp[i + 1]; k++) row.emplace_back(inv[(size_t)t_ci[k]], t_va[k]);
				std::sort(row.begin(), row.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
				int64_t o = p_rp[(size_t)q];
				for (const auto& en : row) {
					p_ci[(size_t)o] = en.first;
					p_va[(size_t)o] = en.second;
					o++;
				}
				p_rp[(size_t)q + 1] = o;
			}
		}
	}
	// Rows beyond one LDS window: T decomposed by the high sites of the species' basis word (lpp_pbseg.h) when T is the hopping matrix of
	// one species in the ascending-word basis -- read off T itself and verified against it entry by entry (pb_seg_plan); otherwise the
	// per-position template of lpp_pbig_kernels.h.  One GPU or the transposition exchange alike: every kernel of a step is position-blind.
	SegPlan SP;
	bool seg = false;
	if (pre) { // the caller made the plan from the model's parameters (pb_chain): there is no host copy of T
		if (W == 0 || cx || tx || n_blk != 1 || c_rp[1] != 0) return fail(LPP_ERR_INVALID, "pb_build: a ready-made plan is for one block beyond the LDS window");
		SP = std::move(*pre);
		seg = pb_seg_lds_bytes(SP.ws, SP.wmax, 1) <= (size_t)160 * 1024 - 64;
		if (!seg) return fail(LPP_ERR_INVALID, "pb_build: the plan exceeds LDS");
	// Taken by itself from 65536 positions per row on: below that the per-position template (80 bytes per position: 3.1 MB at the 38760
	// positions of the 4x5 lattice's 6-electron species) still shares an XCD's L2 with the rows, and k_pb_up_big2 -- compact far lists, fewer
	// lines through L1 -- is faster (the (6,6) sector: 36.8 against 42.7 ms per step; the (7,6) sector, 77520 positions: 91.9 against 90.0).
	// LPP_PB_SEG=0 / 1: never / wherever it applies.
	} else if (W > 0 && !cx && big2 && (getenv("LPP_PB_SEG") ? atoi(getenv("LPP_PB_SEG")) != 0 : n_up >= 65536)) {
		int wcap = (int)std::min<int64_t>(wmax, 8128);
		lpp_status rs = pb_seg_plan(n_up, t_rp, t_ci, t_va, wcap, SP, &seg);
		if (rs != LPP_OK) return rs;
		// One more high site when the two windows of the longest item do not fit beside the tables, or when the cut runs through a ring of
		// the lattice and a segment has more hops than the (5 pairs, 4) kernel instance carries -- the (6, 8) one spills: the (6,6) sector of the
		// 4x5 lattice cut at 4 high sites ran 71 ms per step against 37 with the per-position template
		for (int more = 0; more < 2 && seg && (pb_seg_lds_bytes(SP.ws, SP.wmax, 2) > (size_t)160 * 1024 - 64 || SP.nc_pad > 5); more++) {
			SegPlan SP2;
			bool seg2 = false;
			wcap = std::max(64, SP.wmax - 1);
			rs = pb_seg_plan(n_up, t_rp, t_ci, t_va, wcap, SP2, &seg2);
			if (rs != LPP_OK) return rs;
			if (!seg2 || SP2.s <= SP.s) break;
			SP = std::move(SP2);
		}
		if (seg && pb_seg_lds_bytes(SP.ws, SP.wmax, 2) > (size_t)160 * 1024 - 64) seg = false;
		if (getenv("LPP_VERBOSE"))
			fprintf(stderr, "lpp: segmented in-block form %s (L = %d, n = %d, %d high sites, %zu segments, %zu items of <= %d positions, %d types, %.2f MB)\n", seg ? "taken" : "does not apply",
			        SP.L, SP.n, SP.s, SP.segs.size(), SP.items.size(), SP.wmax, SP.ntypes, (SP.words.size() * 4 + SP.xwords.size() * 4) / 1048576.0);
	}
	const bool permuted = !perm.empty();
	if (getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: product-basis rows %s (pieces %d, exchange %d, blocks %lld of %lld, %lld positions)\n", permuted ? "stored by list length" : "in basis order", W > 0 ? 1 : 0, tx ? 1 : 0, (long long)nblk_loc, (long long)n_blk, (long long)n_up);
	lpp_status rc = LPP_OK;
	if (seg) { // the per-position template is not built at all
		T.G = SP.G;
		for (int g = 0; g < kPbGroupsMax; g++) T.gval[g] = SP.gval[g];
		T.spb = (int)((n_up + 63) / 64);
		T.entries = SP.entries_lo;
		T.slots = SP.slots_lo;
		T.W = W;
	} else
		rc = permuted ? pb_pack_template(n_up, pitch, p_rp.data(), p_ci.data(), p_va.data(), T, ways, W) : pb_pack_template(n_up, pitch, t_rp, t_ci, t_va, T, ways, W);
	if (rc != LPP_OK) return rc;
	if (getenv("LPP_VERBOSE") && !seg) { // list lengths of the packed template: chunks of 4 slots per (slice, group), far slots per slice
		int hist[kPbGroupsMax][9] = { { 0 } }, fh[9] = { 0 };
		for (int j = 0; j < T.spb; j++) {
			for (int g = 0; g < T.G; g++) hist[g][std::min<int>(T.len[(size_t)j * T.G + g], 8)]++;
			if (!T.flen.empty()) fh[std::min<int>((T.flen[(size_t)j] + 3) / 4, 8)]++;
		}
		for (int g = 0; g < T.G; g++) fprintf(stderr, "lpp: template group %d: slices with 0..8+ chunks: %d %d %d %d %d %d %d %d %d\n", g, hist[g][0], hist[g][1], hist[g][2], hist[g][3], hist[g][4], hist[g][5], hist[g][6], hist[g][7], hist[g][8]);
		if (!T.flen.empty()) fprintf(stderr, "lpp: template far slots / 4 per slice 0..8+: %d %d %d %d %d %d %d %d %d; %lld entries in pieces, %lld far\n", fh[0], fh[1], fh[2], fh[3], fh[4], fh[5], fh[6], fh[7], fh[8], (long long)T.entries, (long long)T.far_entries);
	}
	if (permuted) {
		if ((rc = to_device(&B.perm, perm, st)) != LPP_OK) return rc;
		if ((rc = to_device(&B.inv, inv, st)) != LPP_OK) return rc;
	}
	if (seg) {
		if ((rc = to_device(&B.perm, SP.perm, st)) != LPP_OK) return rc;
		if ((rc = to_device(&B.inv, SP.inv, st)) != LPP_OK) return rc;
		if ((rc = to_device((SegItem**)&B.seg_items, SP.items, st)) != LPP_OK) return rc;
		if ((rc = to_device((SegInst**)&B.seg_segs, SP.segs, st)) != LPP_OK) return rc;
		if ((rc = to_device((SegCross**)&B.seg_cross, SP.cross, st)) != LPP_OK) return rc;
		if ((rc = to_device((SegHh**)&B.seg_hh, SP.hh, st)) != LPP_OK) return rc;
		if ((rc = to_device((SegSlice**)&B.seg_slices, SP.slices, st)) != LPP_OK) return rc;
		if ((rc = to_device(&B.seg_tw, SP.words, st)) != LPP_OK) return rc;
		if ((rc = to_device(&B.seg_xw, SP.xwords, st)) != LPP_OK) return rc;
		B.seg = true;
		B.seg_nitems = (int)SP.items.size();
		B.seg_nsegs = (int)SP.segs.size();
		B.seg_ws = SP.ws;
		B.seg_wmax = SP.wmax;
		B.seg_nc = SP.nc_pad;
		B.seg_nh = SP.nh_pad;
		B.seg_pre0 = SP.pre0;
		B.seg_one = pre != nullptr; // one block per workgroup

		B.seg_bytes = (int64_t)(SP.words.size() * 4 + SP.xwords.size() * 4 + SP.slices.size() * 16 + SP.cross.size() * 32 + SP.hh.size() * 16 + SP.segs.size() * 32 + SP.items.size() * 32);
	}
	B.n_up = n_up;
	B.n_blk = n_blk;
	B.pitch = pitch;
	B.tx = tx;
	B.blk0 = blk0;
	B.nblk_loc = nblk_loc;
	B.pitch_dn = pitch_dn;
	B.G = T.G;
	for (int g = 0; g < kPbGroupsMax; g++) B.gval[g] = T.gval[g];
	B.spb = T.spb;
	B.tw_words = (int64_t)T.words.size();
	B.t_entries = T.entries;
	B.t_slots = T.slots;
	const bool pairs = kPbPairs && W == 0; // one-window form: chunk pairs, even look-ahead depths (2,6), (4,4) or (6,4)
	if (pairs && !seg) {
		// chunk pairs: lane l's two words of chunk 2q and of chunk 2q+1 side by side (one 16-byte load); a list of an odd number of chunks ends
		// with half a pair of zero-slot indices; offsets count pairs (k_pb_up)
		std::vector<uint32_t> pw;
		pw.reserve(T.words.size() + T.words.size() / 4);
		const uint32_t zw = T.words.empty() ? 0u : T.words.back(); // the slack behind the lists: (zero slot, zero slot)
		for (size_t i = 0; i < T.off.size(); i++) {
			const size_t src = (size_t)T.off[i] * 128;
			const int nc = T.len[i];
			T.off[i] = (int32_t)(pw.size() / 256);
			for (int q = 0; q < (nc + 1) / 2; q++)
				for (int l = 0; l < 64; l++)
					for (int h = 0; h < 2; h++)
						for (int k = 0; k < 2; k++) pw.push_back(2 * q + h < nc ? T.words[src + (size_t)(2 * q + h) * 128 + (size_t)l * 2 + k] : zw);
		}
		pw.resize(pw.size() + 256 * 4, zw); // slack for the look-ahead loads
		T.words.swap(pw);
		B.tw_words = (int64_t)T.words.size();
	}
	if ((rc = to_device(&B.tw, T.words, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.tw_off, T.off, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.tw_len, T.len, st)) != LPP_OK) return rc;
	B.pre0 = kPbPre;
	if (T.G == 2 && !seg) {
		// look-ahead split of the in-block kernels: the depth pair (3,5), (4,4) or (5,3) that leaves the fewest chunks beyond it
		int64_t best = -1;
		for (int p0 = pairs ? 2 : 3; p0 <= (pairs ? 6 : 5); p0 += pairs ? 2 : 1) {
			int64_t beyond = 0;
			const int p1 = pairs ? pb_depth(2, 1, p0) : 2 * kPbPre - p0;
			for (int j = 0; j < T.spb; j++) beyond += std::max(0, (int)T.len[(size_t)j * 2] - p0) + std::max(0, (int)T.len[(size_t)j * 2 + 1] - p1);
			if (best < 0 || beyond < best || (beyond == best && p0 == kPbPre)) {
				best = beyond;
				B.pre0 = p0;
			}
		}
		if (const char* s = getenv("LPP_PB_PRE0")) B.pre0 = pairs ? std::max(2, std::min(atoi(s), 6)) & ~1 : std::max(3, std::min(atoi(s), 5));
	}
	B.big = W > 0;
	B.big2 = B.big && big2 && (T.G == 1 || T.G == 2 || (T.G == 4 && big2_four)) && pb_big2_lds_bytes((int)W) <= (size_t)160 * 1024 - 64 && (W + kPbZeroSlots) * 8 < 65536;
	B.W = (int)W;
	B.npieces = seg ? B.seg_nitems : W > 0 ? (int)((n_up + W - 1) / W) : 1;
	if (B.big && !seg) {
		B.f_words = (int64_t)T.fwords.size();
		B.f_entries = T.far_entries;
		if ((rc = to_device(&B.fw, T.fwords, st)) != LPP_OK) return rc;
		if ((rc = to_device(&B.f_off, T.foff, st)) != LPP_OK) return rc;
		if ((rc = to_device(&B.f_len, T.flen, st)) != LPP_OK) return rc;
		if (pb_big_lds_bytes(B.W) > (size_t)160 * 1024 - 64) return fail(LPP_ERR_INVALID, "pb_build: piece exceeds LDS");
	}
	// plain off-diagonal copies of T and C (lpp_engine_get_csr walks them; k_pb_down stages C in LDS)
	std::vector<int64_t> tp((size_t)n_up + 1, 0), cp((size_t)n_blk + 1, 0);
	std::vector<int32_t> tc, cc;
	std::vector<double> tv;
	std::vector<uint8_t> ccode;
	const int64_t n_rows_t = cx ? cx->n_c : n_up; // rows of the matrix lpp_engine_get_csr walks
	if (cx) tp.assign((size_t)n_rows_t + 1, 0);
	if (!cx && !t_rp) tp.assign(1, 0); // a ready-made plan (pb_chain): no host T, and lpp_engine_get_csr re-runs the assembler -- nothing to keep
	{
		const int64_t* rp = cx ? cx->t_rp : t_rp;
		const int32_t* ci = cx ? cx->t_ci : t_ci;
		const double* va = cx ? cx->t_va : t_va;
		for (int64_t r = 0; r < n_rows_t && rp; r++) {
			for (int64_t p = rp[r]; p < rp[r + 1]; p++) {
				if (ci[p] == r) continue;
				if (!tc.empty() && (int64_t)tc.size() > tp[(size_t)r] && tc.back() >= ci[p]) return fail(LPP_ERR_INVALID, "pb_build: in-block rows must be sorted by column");
				tc.push_back(ci[p]);
				if (cx) {
					tv.push_back(va[2 * p]);
					tv.push_back(va[2 * p + 1]);
				} else
					tv.push_back(va[p]);
			}
			tp[(size_t)r + 1] = (int64_t)tc.size();
		}
	}
	std::vector<double> cdict; // complex hoppings: the couplings' own dictionary; code 0 = 0 (the padding places of k_pb_down)
	if (cx) cdict.assign(2, 0.0);
	int64_t longest = 1;
	for (int64_t b = 0; b < n_blk; b++) {
		for (int64_t p = c_rp[b]; p < c_rp[b + 1]; p++) {
			if (c_ci[p] == b) continue;
			if (c_ci[p] < 0 || c_ci[p] >= n_blk) return fail(LPP_ERR_INVALID, "pb_build: block coupling out of range");
			if ((int64_t)cc.size() > cp[(size_t)b] && cc.back() >= c_ci[p]) return fail(LPP_ERR_INVALID, "pb_build: block couplings must be sorted");
			uint8_t code = 0;
			if (cx) {
				size_t k = 0;
				for (; k < cdict.size() / 2; k++)
					if (std::memcmp(&cdict[2 * k], &cx->c_va[2 * p], 16) == 0) break;
				if (k == cdict.size() / 2) {
					if (k >= 256) return fail(LPP_ERR_INVALID, "pb_build: more than 256 distinct complex coupling values");
					cdict.push_back(cx->c_va[2 * p]);
					cdict.push_back(cx->c_va[2 * p + 1]);
				}
				code = (uint8_t)k;
			} else {
				code = code_of(dict256, ndict, c_va[p]);
				if (std::memcmp(&dict256[code], &c_va[p], 8) != 0) return fail(LPP_ERR_INVALID, "pb_build: coupling value missing from the dictionary");
			}
			cc.push_back(c_ci[p]);
			ccode.push_back(code);
		}
		cp[(size_t)b + 1] = (int64_t)cc.size();
		longest = std::max(longest, cp[(size_t)b + 1] - cp[(size_t)b]);
	}
	B.c_nnz = (int64_t)cc.size();
	const int64_t longest_list = longest;
	if ((rc = to_device(&B.t_ptr, tp, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.t_col, tc, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.t_val, tv, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.c_ptr, cp, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.c_col, cc, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.c_code, ccode, st)) != LPP_OK) return rc;
	std::vector<double> dict(dict256, dict256 + 256);
	if ((rc = to_device(&B.dict, dict, st)) != LPP_OK) return rc;
	B.ndict = ndict;
	if (cx) {
		cdict.resize(512, 0.0);
		if ((rc = to_device(&B.cdict, cdict, st)) != LPP_OK) return rc;
		B.cplx = true;
		B.n_c = cx->n_c;
	}
	// first CSR entry of every block: a block holds Z_T + n_up*(1 + couplings of the block) entries
	std::vector<int64_t> base((size_t)n_blk + 1, 0);
	const int64_t zt = tp[(size_t)n_rows_t];
	for (int64_t b = 0; b < n_blk; b++) base[(size_t)b + 1] = base[(size_t)b] + zt + n_rows_t * (1 + cp[(size_t)b + 1] - cp[(size_t)b]);
	B.nnz = base[(size_t)n_blk];
	B.nnz_loc = base[(size_t)(blk0 + nblk_loc)] - base[(size_t)blk0];
	if (pre) B.nnz = B.nnz_loc = pre_nnz;
	if ((rc = to_device(&B.blockbase, base, st)) != LPP_OK) return rc;
	// k_pb_down geometry: one workgroup per CU, 8 groups; the couplings of a workgroup's blocks must fit LDS
	int grid = e->num_cus & ~7;
	if (grid < 8) grid = e->num_cus; // fewer than 8 CUs: one group
	const int slots = grid >= 8 ? grid / 8 : grid;
	B.rowcap = (int)std::max<int64_t>(4, (longest_list + 3) & ~(int64_t)3);
	B.ids_per_wg = (int)((n_blk + slots - 1) / slots);
	B.down_grid = grid;
	// the LDS image of a workgroup's coupling lists (150 KB at most): when its blocks' lists do not fit, the workgroup walks its range in
	// rounds of ids_per_round blocks inside every panel (k_pb_down, whole-panel form only); LPP_PB_DOWN_ROUNDS=n forces n rounds (tests)
	B.down_rounds = 1;
	B.ids_per_round = B.ids_per_wg;
	if (!parts) { // (the parts form has an image of its own)
		const int unit = B.ids_per_wg >= 512 ? 64 : 8; // whole sorting windows of `order` (small ranges: whole tasks)
		auto per_round = [&](int r) { return r == 1 ? B.ids_per_wg : (int)((((int64_t)B.ids_per_wg + r - 1) / r + unit - 1) / unit * unit); };
		// Pieces of ~320 blocks also run FASTER than one long range where a workgroup owns 800 blocks and more (measured on the 4x5 lattice,
		// scripts/experiments/README.md "Round 5": 38760 blocks, 1212 per workgroup: 33.2 ms in one round, 28.5 in four; 77520 blocks, 2423 per
		// workgroup: 82.6 ms in the two rounds LDS asks for, 76.8 in eight), and slower below (config 2, 403 per workgroup: 1.30 / 1.50 / 1.69 ms
		// in 1 / 2 / 3 rounds) -- and slower on the same 38760 blocks when the vector is 12 instead of 24 GB (the (6,6) sector: 12.1 ms in one round,
		// 13.5 in four): what the rounds cure grows with the span of addresses a panel touches (one line every 620 KB over 24 GB at the (7,6)
		// sector, whose coupling kernel takes 36 % longer per element than the (6,6) sector's in one round, 17 % in four).  So: vectors from 16 GB on
		int r = 1;
		if (const char* s = getenv("LPP_PB_DOWN_ROUNDS")) r = std::max(1, std::min(atoi(s), 64));
		else if (B.ids_per_wg >= 800 && vec_bytes >= ((size_t)16 << 30)) r = std::min(64, (B.ids_per_wg + 160) / 320);
		while (r < 64 && pb_down_lds_bytes(per_round(r), B.rowcap) > (size_t)150 * 1024) r++;
		while (r > 1 && (int64_t)per_round(r) * (r - 1) >= B.ids_per_wg) r--; // (a last round without blocks: one round less covers the range)
		B.down_rounds = r;
		B.ids_per_round = per_round(r);
	}
	B.down_lds = pb_down_lds_bytes(B.ids_per_round, B.rowcap);
	if (n_blk >= ((int64_t)1 << 24)) return fail(LPP_ERR_INVALID, "pb_build: more than 2^24 - 1 blocks");
	B.parts = parts;
	B.wide = wide;
	if (parts) {
		// per block and part: where the part's entries start in the (ascending) list; per part: the longest list, in whole chunks of 4
		B.nparts = nparts;
		B.part_blocks = (n_blk + nparts - 1) / nparts;
		std::vector<int32_t> pstart((size_t)n_blk * (size_t)(nparts + 1), 0);
		for (int64_t b = 0; b < n_blk; b++) {
			int32_t* ps = pstart.data() + (size_t)b * (size_t)(nparts + 1);
			const int64_t len = cp[(size_t)b + 1] - cp[(size_t)b];
			if (len > 1020) return fail(LPP_ERR_INVALID, "pb_build: coupling list too long");
			int h = 0;
			for (int64_t k = 0; k < len; k++) {
				const int hk = (int)std::min<int64_t>(cc[(size_t)(cp[(size_t)b] + k)] / B.part_blocks, nparts - 1);
				while (h < hk) ps[++h] = (int32_t)k;
			}
			while (h < nparts) ps[++h] = (int32_t)len;
		}
		if ((rc = to_device(&B.c_pstart, pstart, st)) != LPP_OK) return rc;
		// list entries per workgroup (contiguous ranges of ids_per_wg blocks): the largest sizes the LDS image
		int64_t ent_cap = 0;
		for (int64_t lo = 0; lo < n_blk; lo += B.ids_per_wg) ent_cap = std::max(ent_cap, cp[(size_t)std::min<int64_t>(lo + B.ids_per_wg, n_blk)] - cp[(size_t)lo]);
		if (ent_cap > 65000) return fail(LPP_ERR_INVALID, "pb_build: block couplings of a workgroup exceed the 16-bit places of the LDS image");
		B.ent_cap = (int)ent_cap;
		B.down_lds = pb_parts_lds_bytes(B.ids_per_wg, B.ent_cap, nparts);
		constexpr int nw = kPbPartsThreads / 64;
		const int rounds = (((B.ids_per_wg + 7) / 8) + nw - 1) / nw; // tasks of 8 blocks over the waves of a workgroup
		B.maxr = (rounds + 3) / 4 * 4;
		if (rounds > 24) return fail(LPP_ERR_INVALID, "pb_build: too many blocks per workgroup for the register accumulators");
		const int64_t npanels = std::max(pitch, pitch_dn) / 16;
		B.pace_stride = (int)(((npanels + 7) / 8 + 1) * nparts);
		if (grid < 8) B.pace_stride = (int)((npanels + 1) * nparts);
	}
	if (B.down_lds > (size_t)150 * 1024) return fail(LPP_ERR_INVALID, "pb_build: block couplings of a workgroup exceed LDS");
	if (!B.big && pb_up_lds_bytes(pitch, T.spb, T.G) > (size_t)160 * 1024 - 64) return fail(LPP_ERR_INVALID, "pb_build: window + template metadata exceed LDS");
	if ((size_t)nblk_padded * (size_t)(pitch_dn >> 4) >= ((size_t)1 << 32)) return fail(LPP_ERR_INVALID, "pb_build: vector beyond 32-bit line numbers");
	{
		// every workgroup takes its blocks in the order of decreasing list length (tasks of 8 blocks with equal trip counts)
		std::vector<int32_t> order((size_t)n_blk);
		for (int64_t b = 0; b < n_blk; b++) order[(size_t)b] = (int32_t)b;
		// ... inside windows of 64 consecutive blocks: neighbouring blocks gather many of the same lines (hops among the low sites
		// keep the high part of the word -- a run of <= 70 consecutive blocks at config 2), so the tasks that are in flight together
		// share them in L1.  Measured at config 2 (scripts/experiments/r03_order_ab.sh): k_pb_down<RMW> 1.711 ms sorted over the
		// whole range of ~400 blocks, 1.655-1.662 ms with windows of 48-64, 1.73 ms with windows of 32 (tasks of unequal lists)
		int64_t order_window = 64;
		if (const char* s = getenv("LPP_PB_ORDER_WINDOW")) order_window = std::max(0, atoi(s)) / 16 * 16;
		for (int sl = 0; sl < slots; sl++) {
			const int64_t lo = std::min<int64_t>((int64_t)sl * B.ids_per_wg, n_blk), hi = std::min<int64_t>(lo + B.ids_per_wg, n_blk);
			if (parts) {
				// tasks of 8 consecutive blocks run as long as their longest list IN EVERY PART: blocks with the same profile of
				// part lengths belong together (measured on N_dn = 38760 in 3 parts: 1.11 instead of 1.19 gathers issued per entry)
				const int64_t pbk = (n_blk + nparts - 1) / nparts;
				auto profile = [&](int32_t b, int* out) {
					for (int q = 0; q < nparts; q++) out[q] = 0;
					for (int64_t p = cp[(size_t)b]; p < cp[(size_t)b + 1]; p++) out[std::min<int64_t>(cc[(size_t)p] / pbk, nparts - 1)]++;
				};
				std::stable_sort(order.begin() + lo, order.begin() + hi, [&](int32_t x, int32_t y) {
					int px[kPbMaxParts], py[kPbMaxParts];
					profile(x, px);
					profile(y, py);
					for (int q = 0; q < nparts; q++)
						if (px[q] != py[q]) return px[q] > py[q];
					return false;
				});
			} else {
				// sorted inside windows of `ow` consecutive blocks (a multiple of 16; 0: the whole range at once)
				const int64_t ow = order_window > 0 ? order_window : hi - lo;
				for (int64_t w0 = lo; w0 < hi; w0 += ow)
					std::stable_sort(order.begin() + w0, order.begin() + std::min(w0 + ow, hi), [&](int32_t x, int32_t y) { return cp[(size_t)x + 1] - cp[(size_t)x] > cp[(size_t)y + 1] - cp[(size_t)y]; });
			}
		}
		if ((rc = to_device(&B.order, order, st)) != LPP_OK) return rc;
	}
	if (B.down_rounds > 1 && !(getenv("LPP_PB_DOWN_IMAGE") && atoi(getenv("LPP_PB_DOWN_IMAGE")) == 0)) {
		// the LDS images of all (workgroup, round) pairs, made once: a round then starts with a copy instead of a walk through the lists
		// ((7,6) sector of the 4x5 lattice forced into two rounds: coupling kernel 33.2 ms in one round, 43.0 ms rebuilding)
		const size_t bytes = pb_down_lds_bytes(B.ids_per_round, B.rowcap) * (size_t)slots * (size_t)B.down_rounds;
		HIP_TRY_MEM(hipMalloc(&B.down_image, bytes));
		PbDownArgs d = {};
		d.n_blk = n_blk;
		d.ids_per_wg = B.ids_per_wg;
		d.rounds = B.down_rounds;
		d.ids_per_round = B.ids_per_round;
		d.rowcap = B.rowcap;
		d.c_ptr = B.c_ptr;
		d.c_col = B.c_col;
		d.c_code = B.c_code;
		d.order = B.order;
		k_pb_down_image<<<slots * B.down_rounds, 256, 0, st>>>(d, (uint4*)B.down_image);
	}
	if (!(getenv("LPP_PB_PACE") && atoi(getenv("LPP_PB_PACE")) == 0))
		HIP_TRY_MEM(hipMalloc(&B.pace, sizeof(int) * 8 * (parts ? (size_t)B.pace_stride : (size_t)(std::max(pitch, pitch_dn) / 16))));
	// the two parts of a product (padding stays zero) and the carried scalar; with the transposition exchange the couplings'
	// part is written straight into the exchange buffer
	const size_t loc = (size_t)std::max<int64_t>(nblk_loc, 1) * (size_t)pitch;
	if (!tx) {
		HIP_TRY_MEM(hipMalloc(&B.z, sizeof(double) * loc));
		HIP_TRY(hipMemsetAsync(B.z, 0, sizeof(double) * loc, st));
	}
	HIP_TRY_MEM(hipMalloc(&B.u, sizeof(double) * loc));
	HIP_TRY(hipMemsetAsync(B.u, 0, sizeof(double) * loc, st));
	HIP_TRY_MEM(hipMalloc(&B.xy, sizeof(double) * 2));
	HIP_TRY(hipMemsetAsync(B.xy, 0, sizeof(double) * 2, st));
	HIP_TRY_MEM(hipMalloc(&B.dcode, loc));
	HIP_TRY(hipMemsetAsync(B.dcode, 0, loc, st));
	HIP_TRY(hipStreamSynchronize(st));
	e->pitch = cx ? pitch / 2 : pitch; // in vector elements
	e->pitch_rows = cx ? cx->n_c : n_up;
	e->pitch_blocks = nblk_loc;
	B.active = true;
	return LPP_OK;
}

// x = beta x + alpha H y (EpiScale semantics of the other product kernels) as two independent kernels that only read y,
//   k_pb_down  z = alpha C y            (+ partials of Re<y|z>)
//   k_pb_up    u = alpha (T y + D y)    (+ partials of Re<y|u>, stored in front of the first kernel's)
// and a streaming pass x = beta x + u + z (k_pb_combine).  defer_combine: the caller runs that pass itself, folded into its own
// pass over x (pb_combine_axpy of the scale-free recurrence).  Returns the number of partials written; their sum is
// Re<y | u + z>, to which the caller adds beta Re<y | x_old> (pb.xy, left by the previous combine pass).
constexpr int kPreLo = kPbPairs ? 2 : 3, kPreHi = kPbPairs ? 6 : 5; // the unequal look-ahead splits k_pb_up is instantiated for
template <bool DOT> static void launch_up(const PbState& B, const PbUpArgs& u, int nb, size_t lds, hipStream_t st)
{
	const int gt = B.G <= 2 ? B.G : 0;
#define LPP_PB_UP(GT_)                                                                                                \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_up<DOT, GT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		k_pb_up<DOT, GT_><<<nb, kPbUpThreads, lds, st>>>(u);                                                            \
	} while (0)
#define LPP_PB_UP2(PRE_)                                                                                              \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_up<DOT, 2, false, PRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		k_pb_up<DOT, 2, false, PRE_><<<nb, kPbUpThreads, lds, st>>>(u);                                                 \
	} while (0)
	if (gt == 1) LPP_PB_UP(1);
	else if (gt == 2 && B.pre0 == kPreLo) LPP_PB_UP2(kPreLo); // look-ahead split of the two value groups (pb_build)
	else if (gt == 2 && B.pre0 == kPreHi) LPP_PB_UP2(kPreHi);
	else if (gt == 2) LPP_PB_UP(2);
	else LPP_PB_UP(0);
#undef LPP_PB_UP2
#undef LPP_PB_UP
}


// ---- rows beyond one LDS window / vectors beyond 4 GiB (lpp_pbig_kernels.h) -------------------------------------------
static int big_grid(const lpp_engine* e, int64_t cnt)
{
	const PbState& B = e->pb;
	if (B.big2 || B.seg) { // one workgroup per CU, items = (pair of blocks of one XCD, piece)
		int nb = (int)std::max<int64_t>(1, std::min<int64_t>((B.seg_one ? cnt : (cnt + 1) / 2) * B.npieces, (int64_t)e->num_cus));
		if (nb >= 8) nb &= ~7;
		return nb;
	}
	const size_t lds = pb_big_lds_bytes(B.W);
	const int per_cu = std::max(1, std::min(2, (int)(((size_t)160 * 1024) / (lds + 512))));
	int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cnt * B.npieces, (int64_t)e->num_cus * per_cu));
	if (nb >= 8) nb &= ~7;
	return nb;
}

// in-block part by pieces: `cnt` blocks from `y` (pitched), result to `u`; returns the number of partials written
static int launch_up_big(lpp_engine* e, const double* y, double* u, const uint8_t* dcode, int64_t cnt, double* partial, const EpiScale& sc, hipStream_t st)
{
	const PbState& B = e->pb;
	PbUpBigArgs a;
	a.tw = B.tw;
	a.tw_off = B.tw_off;
	a.tw_len = B.tw_len;
	a.fw = B.fw;
	a.f_off = B.f_off;
	a.f_len = B.f_len;
	a.G = B.G;
	for (int g = 0; g <= kPbMaxGroups; g++) a.gval[g] = g < B.G ? B.gval[g] : 0.0;
	a.dict = B.dict;
	a.dcode = dcode;
	a.n_up = B.n_up;
	a.pitch = B.pitch;
	a.n_blk = cnt;
	a.W = B.W;
	a.npieces = B.npieces;
	a.y = y;
	a.u = u;
	a.partial = partial;
	a.sc = sc;
	const int nb = big_grid(e, cnt);
	if (B.seg) {
		PbSegArgs g;
		g.items = (const SegItem*)B.seg_items;
		g.nitems = B.seg_nitems;
		g.cross = (const SegCross*)B.seg_cross;
		g.hh = (const SegHh*)B.seg_hh;
		g.slices = (const SegSlice*)B.seg_slices;
		g.tw = B.seg_tw;
		g.xw = B.seg_xw;
		g.G = B.G;
		g.gval[0] = B.gval[0];
		g.gval[1] = B.G > 1 ? B.gval[1] : 0.0;
		g.dict = B.dict;
		g.dcode = dcode;
		g.n_up = B.n_up;
		g.pitch = B.pitch;
		g.n_blk = cnt;
		g.ws = B.seg_ws;
		g.wmax = B.seg_wmax;
		g.y = y;
		g.u = u;
		g.partial = partial;
		g.sc = sc;
		g.flat = B.seg_one ? 1 : 0;
		const size_t lds = pb_seg_lds_bytes(B.seg_ws, B.seg_wmax, B.seg_one ? 1 : 2);
#define LPP_PB_SEG(DOT_, GT_, P0_, NC_, NH_)                                                                            \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_up_seg<DOT_, GT_, P0_, 4, NC_, NH_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		k_pb_up_seg<DOT_, GT_, P0_, 4, NC_, NH_><<<nb, kSegThreads, lds, st>>>(g);                                      \
	} while (0)
#define LPP_PB_SEG1(DOT_, GT_, P0_, NC_, NH_)                                                                           \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_up_seg<DOT_, GT_, P0_, 4, NC_, NH_, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		k_pb_up_seg<DOT_, GT_, P0_, 4, NC_, NH_, 1><<<nb, kSegThreads, lds, st>>>(g);                                   \
	} while (0)
#define LPP_PB_SEG_N(DOT_, GT_, P0_)                                                                                   \
	do {                                                                                                              \
		if (B.seg_one && B.seg_nh == 12 && B.seg_nc == 1) LPP_PB_SEG1(DOT_, GT_, P0_, 1, 12);                           \
		else if (B.seg_one && B.seg_nh == 12) LPP_PB_SEG1(DOT_, GT_, P0_, 2, 12);                                       \
		else if (B.seg_one && B.seg_nh == 16 && B.seg_nc == 1) LPP_PB_SEG1(DOT_, GT_, P0_, 1, 16);                      \
		else if (B.seg_one && B.seg_nh == 16) LPP_PB_SEG1(DOT_, GT_, P0_, 2, 16);                                       \
		else if (B.seg_one) LPP_PB_SEG1(DOT_, GT_, P0_, 2, 2);                                                          \
		else if (B.seg_nc == 2) LPP_PB_SEG(DOT_, GT_, P0_, 2, 2);                                                            \
		else if (B.seg_nc == 5) LPP_PB_SEG(DOT_, GT_, P0_, 5, 4);                                                       \
		else LPP_PB_SEG(DOT_, GT_, P0_, 6, 8);                                                                          \
	} while (0)
		// cross / high-high hops per segment: the instance the plan padded its lists for; chunks of value group 0 requested ahead: its choice
		if (partial) {
			if (B.G == 1) LPP_PB_SEG_N(true, 1, 4);
			else if (B.seg_pre0 == 2) LPP_PB_SEG_N(true, 2, 2);
			else LPP_PB_SEG_N(true, 2, 4);
		} else {
			if (B.G == 1) LPP_PB_SEG_N(false, 1, 4);
			else if (B.seg_pre0 == 2) LPP_PB_SEG_N(false, 2, 2);
			else LPP_PB_SEG_N(false, 2, 4);
		}
#undef LPP_PB_SEG_N
#undef LPP_PB_SEG1
#undef LPP_PB_SEG
		return partial ? nb : 0;
	}
	if (B.big2) {
		const size_t lds2 = pb_big2_lds_bytes(B.W);
#define LPP_PB_BIG2(DOT_, GT_, PRE_)                                                                                   \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_up_big2<DOT_, GT_, PRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); \
		k_pb_up_big2<DOT_, GT_, PRE_><<<nb, kPbBig2Threads, lds2, st>>>(a);                                             \
	} while (0)
		if (partial) {
			if (B.G == 1) LPP_PB_BIG2(true, 1, 4);
			else if (B.G == 4) LPP_PB_BIG2(true, 4, 2); // two chunks of each of the four groups ahead (three spill: 5.00 against 4.74 ms at 1.47e8 complex states)
			else if (B.pre0 == 3) LPP_PB_BIG2(true, 2, 3);
			else if (B.pre0 == 5) LPP_PB_BIG2(true, 2, 5);
			else LPP_PB_BIG2(true, 2, 4);
		} else {
			if (B.G == 1) LPP_PB_BIG2(false, 1, 4);
			else if (B.G == 4) LPP_PB_BIG2(false, 4, 2);
			else if (B.pre0 == 3) LPP_PB_BIG2(false, 2, 3);
			else if (B.pre0 == 5) LPP_PB_BIG2(false, 2, 5);
			else LPP_PB_BIG2(false, 2, 4);
		}
#undef LPP_PB_BIG2
		return partial ? nb : 0;
	}
	const size_t lds = pb_big_lds_bytes(B.W);
	const int gt = B.G <= 2 || B.G == 4 ? B.G : 0;
#define LPP_PB_BIG(DOT_, GT_, PRE_)                                                                                    \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_up_big<DOT_, GT_, PRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
		k_pb_up_big<DOT_, GT_, PRE_><<<nb, kPbBigThreads, lds, st>>>(a);                                                \
	} while (0)
	if (partial) {
		if (gt == 1) LPP_PB_BIG(true, 1, kBigPre);
		else if (gt == 2) LPP_PB_BIG(true, 2, kBigPre);
		else if (gt == 4) LPP_PB_BIG(true, 4, 3);
		else LPP_PB_BIG(true, 0, kBigPre);
	} else {
		if (gt == 1) LPP_PB_BIG(false, 1, kBigPre);
		else if (gt == 2) LPP_PB_BIG(false, 2, kBigPre);
		else if (gt == 4) LPP_PB_BIG(false, 4, 3);
		else LPP_PB_BIG(false, 0, kBigPre);
	}
#undef LPP_PB_BIG
	return partial ? nb : 0;
}

// The plain coupling kernel with the tasks of a step handed out by the LDS counter and NO pacing between the workgroups of a group (k_pb_down<.., PF>
// with pace == null: the barrier per step keeps a workgroup together, the counter keeps its waves on neighbouring tasks): faster where a panel of all blocks
// fits an XCD's L2 with room to spare -- complex couplings at 11440 blocks 2.27 -> 2.08 ms, the 3x6 lattice's (6,6) sector (18564 blocks) 2.72 -> 2.56 ms -- and
// slower where it does not (38760 blocks, 4.96 MB: 28.4 -> 35.8 ms; with pacing on top of the counter 28.5): up to 3 MB per panel.  LPP_PB_DOWN_TASKS=0 / 1
static bool pb_down_tasks(const PbState& B, int64_t n_blk)
{
	if (const char* s = getenv("LPP_PB_DOWN_TASKS")) return atoi(s) != 0;
	(void)B;
	return (size_t)n_blk * 128 <= (size_t)3 << 20;
}

static void pb_launch_down_plain(const PbState& B, PbDownArgs& d, hipStream_t st)
{
	const bool tasks = pb_down_tasks(B, d.n_blk);
	if (tasks) d.pace = nullptr;
	if (d.pace) (void)hipMemsetAsync(d.pace, 0, sizeof(int) * 8 * (size_t)d.npanels, st);
#define LPP_PB_DOWN_PLAIN(WIDE_, CPLX_, PF_)                                                                           \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_down<1024, false, WIDE_, CPLX_, PF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B.down_lds); \
		k_pb_down<1024, false, WIDE_, CPLX_, PF_><<<B.down_grid, 1024, B.down_lds, st>>>(d);                            \
	} while (0)
	if (B.cplx && B.wide) {
		if (tasks) LPP_PB_DOWN_PLAIN(true, true, true);
		else LPP_PB_DOWN_PLAIN(true, true, false);
	} else if (B.cplx) {
		if (tasks) LPP_PB_DOWN_PLAIN(false, true, true);
		else LPP_PB_DOWN_PLAIN(false, true, false);
	} else if (B.wide) {
		if (tasks) LPP_PB_DOWN_PLAIN(true, false, true);
		else LPP_PB_DOWN_PLAIN(true, false, false);
	} else {
		if (tasks) LPP_PB_DOWN_PLAIN(false, false, true);
		else LPP_PB_DOWN_PLAIN(false, false, false);
	}
#undef LPP_PB_DOWN_PLAIN
}

// block couplings over parts of the source range: z = alpha C y on rows of `pitch` positions; returns the number of partials written
static int launch_down_parts(lpp_engine* e, const double* y, double* z, int64_t pitch, double* partial, const EpiScale& sc, hipStream_t st)
{
	const PbState& B = e->pb;
	PbDownPartsArgs d;
	d.pitch = pitch;
	d.n_blk = B.n_blk;
	d.npanels = (int)(pitch / 16);
	d.ids_per_wg = B.ids_per_wg;
	d.nparts = B.nparts;
	d.ent_cap = B.ent_cap;
	d.c_ptr = B.c_ptr;
	d.c_col = B.c_col;
	d.c_code = B.c_code;
	d.c_pstart = B.c_pstart;
	d.order = B.order;
	d.dict = B.dict;
	d.y = y;
	d.z = z;
	d.partial = partial;
	d.sc = sc;
	d.pace = B.pace;
	d.pace_stride = B.pace_stride;
	if (d.pace) (void)hipMemsetAsync(d.pace, 0, sizeof(int) * 8 * (size_t)B.pace_stride, st);
#define LPP_PB_PARTS(R_)                                                                                               \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_down_parts<kPbPartsThreads, R_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B.down_lds); \
		k_pb_down_parts<kPbPartsThreads, R_><<<B.down_grid, kPbPartsThreads, B.down_lds, st>>>(d);                      \
	} while (0)
	if (B.maxr <= 4) LPP_PB_PARTS(4);
	else if (B.maxr <= 8) LPP_PB_PARTS(8);
	else if (B.maxr <= 12) LPP_PB_PARTS(12);
	else if (B.maxr <= 16) LPP_PB_PARTS(16);
	else if (B.maxr <= 20) LPP_PB_PARTS(20);
	else LPP_PB_PARTS(24);
#undef LPP_PB_PARTS
	return partial ? B.down_grid : 0;
}

// 8192 blocks x 4 elements in flight: 5.2 TB/s for the 4-read 1-write mix (4.7 with 2048 x 2), scripts/experiments/calib_combine.hip
static int combine_blocks(int64_t n2) { return (int)std::max<int64_t>(1, std::min<int64_t>((n2 + 4 * kBlock - 1) / (4 * kBlock), 8192)); }

int pb_launch(lpp_engine* e, const void* y, void* x, double* partial, const EpiScale& sc, bool defer_combine)
{
	PbState& B = e->pb;
	hipStream_t st = e->stream;
	const int nb = B.big ? big_grid(e, B.n_blk) : (int)std::max<int64_t>(1, std::min<int64_t>(B.n_blk, (int64_t)e->num_cus));
	double* const want_dot = partial;
	if (!defer_combine) partial = nullptr; // the combine pass below forms Re<y|x> of the finished x itself
	int np = partial ? nb : 0;
	// (the two kernels only read y; side by side on the same CUs they ran 4.95 instead of 2.69 ms at config 2 -- the in-block kernel's
	// traffic evicts the panel from L2 -- so they run one after the other: scripts/experiments/README.md)
	const bool both = B.c_nnz > 0;
	hipStream_t sd = st;
	if (both && B.parts) {
		const int n = launch_down_parts(e, (const double*)y, B.z, B.pitch, partial ? partial + nb : nullptr, sc, sd);
		if (partial) np += n;
	} else if (both) {
		PbDownArgs d = {};
		d.pitch = B.pitch;
		d.n_blk = B.n_blk;
		d.npanels = (int)(B.pitch / 16);
		d.ids_per_wg = B.ids_per_wg;
	d.rounds = B.down_rounds;
	d.ids_per_round = B.ids_per_round;
	d.image = (const uint4*)B.down_image;
		d.rowcap = B.rowcap;
		d.c_ptr = B.c_ptr;
		d.c_col = B.c_col;
		d.c_code = B.c_code;
		d.dict = B.dict;
		d.y = (const double*)y;
		d.z = B.z;
		d.u_in = nullptr;
		d.shift = nullptr;
		d.partial = partial ? partial + nb : nullptr;
		d.sc = sc;
		d.pace = B.pace;
		d.order = B.order;
		d.u_has_beta = 0;
		d.cdict = (const double2*)B.cdict;
		pb_launch_down_plain(B, d, sd);
		if (partial) np += B.down_grid;
	}
	if (B.big) {
		launch_up_big(e, (const double*)y, B.u, B.dcode, B.n_blk, partial, sc, st); // nb partials, in front of the couplings'
	} else {
	PbUpArgs u = {};
	u.tw = B.tw;
	u.tw_off = B.tw_off;
	u.tw_len = B.tw_len;
	u.G = B.G;
	for (int g = 0; g < kPbMaxGroups; g++) u.gval[g] = B.gval[g];
	u.dict = B.dict;
	u.dcode = B.dcode;
	u.n_up = B.n_up;
	u.pitch = B.pitch;
	u.n_blk = B.n_blk;
	u.spb = B.spb;
	u.y = (const double*)y;
	u.u = B.u;
	u.partial = partial;
	u.sc = sc;
	u.wbuf = u.ybuf = nullptr;
	u.g_a = u.g_b2 = nullptr;
	const size_t lds = pb_up_lds_bytes(B.pitch, B.spb, B.G);
	if (partial) launch_up<true>(B, u, nb, lds, st);
	else launch_up<false>(B, u, nb, lds, st);
	}
	if (!defer_combine) {
		PbCombineArgs c;
		c.n2 = (B.n_blk * B.pitch) >> 1;
		c.x = (double2*)x;
		c.y = (const double2*)y;
		c.u = (const double2*)B.u;
		c.z = B.c_nnz > 0 ? (const double2*)B.z : nullptr;
		c.sc = sc;
		c.a_ptr = nullptr;
		c.b2_prev = nullptr;
		c.d = (const double2*)B.dval;
		c.partial_dq = nullptr;
		const int nbc = combine_blocks(c.n2);
		c.partial_nrm = want_dot ? want_dot + nbc : nullptr; // |x|^2 partials: not used by these callers
		c.partial_xy = want_dot;
		k_pb_combine<<<nbc, kBlock, 0, st>>>(c);
		if (want_dot) np = nbc;
	}
	return np;
}

// ---- several GPUs, transposition exchange -----------------------------------------------------------------------------
// Per step and rank (one_step):  pack -> all-to-all #1 || pb_tx_up(first half of the own blocks) -> pb_tx_down on the received
// transposed slice -> all-to-all #2 || pb_tx_up(second half) -> pb_tx_unpack_combine.  Both parts of the product are the
// single-GPU kernels: the in-block kernel on the rank's own blocks, the panel-major coupling kernel on the transposed slice
// (all N_down blocks, rows = this rank's up-index range: the same C (x) 1 structure with narrower rows).
static void fill_up_args(const PbState& B, PbUpArgs& u)
{
	u.tw = B.tw;
	u.tw_off = B.tw_off;
	u.tw_len = B.tw_len;
	u.G = B.G;
	for (int g = 0; g < kPbMaxGroups; g++) u.gval[g] = B.gval[g];
	u.dict = B.dict;
	u.n_up = B.n_up;
	u.pitch = B.pitch;
	u.spb = B.spb;
	u.partial = nullptr;
	u.wbuf = u.ybuf = nullptr;
	u.g_a = u.g_b2 = nullptr;
}

void pb_tx_up(lpp_engine* e, const void* y, const EpiScale& sc, int64_t b0, int64_t cnt)
{
	PbState& B = e->pb;
	if (cnt <= 0) return;
	if (B.big) {
		launch_up_big(e, (const double*)y + b0 * B.pitch, B.u + b0 * B.pitch, B.dcode + b0 * B.pitch, cnt, nullptr, sc, e->stream);
		return;
	}
	PbUpArgs u = {};
	fill_up_args(B, u);
	u.dcode = B.dcode + b0 * B.pitch;
	u.n_blk = cnt;
	u.y = (const double*)y + b0 * B.pitch;
	u.u = B.u + b0 * B.pitch;
	u.sc = sc;
	const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(cnt, (int64_t)e->num_cus));
	launch_up<false>(B, u, nb, pb_up_lds_bytes(B.pitch, B.spb, B.G), e->stream);
}

void pb_tx_down(lpp_engine* e, const void* gath, void* send2, const EpiScale& sc)
{
	PbState& B = e->pb;
	if (B.parts) {
		launch_down_parts(e, (const double*)gath, (double*)send2, B.pitch_dn, nullptr, sc, e->stream);
		return;
	}
	PbDownArgs d = {};
	d.pitch = B.pitch_dn;
	d.n_blk = B.n_blk;
	d.npanels = (int)(B.pitch_dn / 16);
	d.ids_per_wg = B.ids_per_wg;
	d.rounds = B.down_rounds;
	d.ids_per_round = B.ids_per_round;
	d.image = (const uint4*)B.down_image;
	d.rowcap = B.rowcap;
	d.c_ptr = B.c_ptr;
	d.c_col = B.c_col;
	d.c_code = B.c_code;
	d.dict = B.dict;
	d.y = (const double*)gath;
	d.z = (double*)send2;
	d.u_in = nullptr;
	d.shift = nullptr;
	d.partial = nullptr;
	d.sc = sc;
	d.pace = B.pace;
	d.order = B.order;
	d.u_has_beta = 0;
	pb_launch_down_plain(B, d, e->stream);
}

int pb_tx_unpack_combine(lpp_engine* e, void* x, const void* y, const void* recv2, const EpiScale& sc, int64_t chunk, double* partial, const double* shift)
{
	const PbState& B = e->pb;
	const int64_t n2 = (B.nblk_loc * B.pitch) >> 1;
	const int nb = combine_blocks(n2);
	k_pb_unpack_combine<<<nb, kBlock, 0, e->stream>>>((double2*)x, (const double2*)y, (const double2*)B.u, (const double2*)recv2, B.nblk_loc, B.pitch >> 1,
	                                                  B.pitch_dn >> 1, chunk >> 1, sc, partial, shift, (const double2*)B.dval);
	return nb;
}

// ---- the chained scale-free step (k_pb_up<KC>, k_pb_down<RMW>) -------------------------------------------------
bool pb_chain_ok(const lpp_engine* e)
{
	const PbState& B = e->pb;
	if (getenv("LPP_PB_CHAIN") && atoi(getenv("LPP_PB_CHAIN")) == 0) return false;
	return B.active && !B.tx && !B.big && !B.parts && !B.wide && !B.dval && B.down_rounds == 1 && B.c_nnz > 0 && B.G >= 1 && B.G <= kPbMaxGroups; // more than two value groups (complex hoppings realified, t-t' models): the any-number-of-groups path of k_pb_up
}

template <int GT, int PRE0 = kPbPre> static void launch_up_chain(const PbUpArgs& u, int nb, size_t lds, hipStream_t st)
{
	(void)hipFuncSetAttribute((const void*)k_pb_up<false, GT, true, PRE0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
	k_pb_up<false, GT, true, PRE0><<<nb, kPbUpThreads, lds, st>>>(u);
}

// One scale-free Lanczos step in two launches.  In: w (= w_{j-1}, or r_j itself when g_a is null) and y (= r_{j-1}).
// Out: w holds r_j, y holds w_j = alpha H r_j + beta r_{j-1} (the in-block part passes through pb.u); partial holds pairs (Re<r_j|w_j>, |w_j|^2), their number is returned.
int pb_launch_chain(lpp_engine* e, void* w, void* y, double* partial, const EpiScale& sc, const double* g_a, const double* g_b2, const double* shift)
{
	PbState& B = e->pb;
	hipStream_t st = e->stream;
	const int nb = (int)std::max<int64_t>(1, std::min<int64_t>(B.n_blk, (int64_t)e->num_cus));
	PbUpArgs u = {};
	u.tw = B.tw;
	u.tw_off = B.tw_off;
	u.tw_len = B.tw_len;
	u.G = B.G;
	for (int g = 0; g < kPbMaxGroups; g++) u.gval[g] = B.gval[g];
	u.dict = B.dict;
	u.dcode = B.dcode;
	u.n_up = B.n_up;
	u.pitch = B.pitch;
	u.n_blk = B.n_blk;
	u.spb = B.spb;
	u.y = nullptr;
	u.u = B.u;
	u.partial = nullptr;
	u.sc = sc;
	u.wbuf = (double*)w;
	u.ybuf = (double*)y;
	u.g_a = g_a;
	u.g_b2 = g_b2;
	{
		// beta r_{j-1} rides in u (k_pb_up<CHAIN>), so the coupling kernel reads one stream less
		const size_t lds = pb_up_lds_bytes(B.pitch, B.spb, B.G);
		if (B.G == 1) launch_up_chain<1>(u, nb, lds, st);
		else if (B.G > 2) launch_up_chain<0>(u, nb, lds, st);
		else if (B.pre0 == kPreLo) launch_up_chain<2, kPreLo>(u, nb, lds, st);
		else if (B.pre0 == kPreHi) launch_up_chain<2, kPreHi>(u, nb, lds, st);
		else launch_up_chain<2>(u, nb, lds, st);
	}
	PbDownArgs d = {};
	d.pitch = B.pitch;
	d.n_blk = B.n_blk;
	d.npanels = (int)(B.pitch / 16);
	d.ids_per_wg = B.ids_per_wg;
	d.rounds = B.down_rounds;
	d.ids_per_round = B.ids_per_round;
	d.image = (const uint4*)B.down_image;
	d.rowcap = B.rowcap;
	d.c_ptr = B.c_ptr;
	d.c_col = B.c_col;
	d.c_code = B.c_code;
	d.dict = B.dict;
	d.y = (const double*)w;
	d.z = (double*)y;
	d.u_in = B.u;
	d.shift = shift;
	d.partial = partial;
	d.sc = sc;
	d.pace = B.pace;
	d.order = B.order;
	d.u_has_beta = 1; // the in-block kernel has put beta r_{j-1} into u
	d.cdict = (const double2*)B.cdict;
	// the last wave touches the u lines ahead of the tasks and the tasks are handed out by a counter (k_pb_down<..., PF>): config 2 3.03 -> 2.88 ms
	// per step; LPP_PB_DOWN_PF=0: every wave its fixed share of the tasks, no touching
	static const bool pf = !(getenv("LPP_PB_DOWN_PF") && atoi(getenv("LPP_PB_DOWN_PF")) == 0);
	d.pf_lead = 2;
	// ... and with the tasks handed out by a counter and a barrier per panel the workgroups of a group stay together by themselves: the bounded pacing
	// (one atomic, one poll and one more barrier per panel, a memset per launch) only costs here -- config 2 1.51 -> 1.41 ms, the 4x4 lattice's (7,7) sector
	// 1.19 -> 1.08, complex hoppings at (6,6) 1.25 -> 1.09 ms; the plain kernel needs it (3x6 lattice, (6,6): 2.7 ms with, 6.1 without).  LPP_PB_CHAIN_PACE=1: keep it
	static const bool chain_pace = getenv("LPP_PB_CHAIN_PACE") && atoi(getenv("LPP_PB_CHAIN_PACE")) != 0;
	if (pf && !chain_pace && (size_t)B.n_blk * 128 <= (size_t)3 << 20) d.pace = nullptr; // (a panel beyond an XCD's L2 keeps it: see pb_down_tasks)
	if (d.pace) (void)hipMemsetAsync(d.pace, 0, sizeof(int) * 8 * (size_t)d.npanels, st);
#define LPP_PB_DOWN_RMW(CPLX_, PF_)                                                                                    \
	do {                                                                                                              \
		(void)hipFuncSetAttribute((const void*)k_pb_down<1024, true, false, CPLX_, PF_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)B.down_lds); \
		k_pb_down<1024, true, false, CPLX_, PF_><<<B.down_grid, 1024, B.down_lds, st>>>(d);                             \
	} while (0)
	if (B.cplx && pf) LPP_PB_DOWN_RMW(true, true);
	else if (B.cplx) LPP_PB_DOWN_RMW(true, false);
	else if (pf) LPP_PB_DOWN_RMW(false, true);
	else LPP_PB_DOWN_RMW(false, false);
#undef LPP_PB_DOWN_RMW
	return B.down_grid;
}

// leave the chained form: run the pending pass y = y - g x and restore the carried <y | x_old>
void pb_materialise(lpp_engine* e, void* y, const void* x, const double* g_a, const double* g_b2, double* partial)
{
	const PbState& B = e->pb;
	const int64_t n2 = (B.n_blk * B.pitch) >> 1;
	const int nb = combine_blocks(n2);
	k_pb_materialise<<<nb, kBlock, 0, e->stream>>>((double2*)y, (const double2*)x, g_a, g_b2, n2, partial);
	k_reduce_final<<<1, kBlock, 0, e->stream>>>(partial, nb, 1, 1, B.xy);
}

int pb_combine_axpy(lpp_engine* e, void* x, const void* y, const EpiScale& sc, const double* a_ptr, const double* b2_prev, double* partial)
{
	const PbState& B = e->pb;
	PbCombineArgs c;
	c.n2 = (B.n_blk * B.pitch) >> 1;
	c.x = (double2*)x;
	c.y = (const double2*)y;
	c.u = (const double2*)B.u;
	c.z = B.c_nnz > 0 ? (const double2*)B.z : nullptr;
	c.sc = sc;
	c.a_ptr = a_ptr;
	c.b2_prev = b2_prev;
	const int nb = combine_blocks(c.n2);
	c.partial_nrm = partial;
	c.partial_xy = partial + nb;
	c.d = (const double2*)B.dval;
	c.partial_dq = B.dval ? partial + 2 * nb : nullptr;
	k_pb_combine<<<nb, kBlock, 0, e->stream>>>(c);
	k_reduce_final<<<1, kBlock, 0, e->stream>>>(partial + nb, nb, 1, 1, B.xy); // next step's <y | x_old>
	if (B.dval) k_reduce_final<<<1, kBlock, 0, e->stream>>>(partial + 2 * nb, nb, 1, 1, B.xy + 1); // and its <y | D y>
	return nb;
}

// ---- a CSR that was handed over (lpp_engine_set_csr / _set_csr_device) -------------------------------------------------
// The reference hands its matrix over as a CSR (DefaultSymmetry.h:54-57 -> InternalProductStored.h:116).  When that CSR is of
// the product-basis form -- basis block n_up (the caller's hint or the detected one), every block's in-block part equal to
// block 0's (= T), every leaving entry a position-preserving block coupling (= C), a stored diagonal in every row (= D) -- it
// is taken into the same layout device assembly builds: T from block 0, C from the first row of every block, D from the
// diagonals, then EVERY row of the CSR is compared with the row (T, C, D) stand for, entry by entry and bit by bit
// (k_pb_csr_verify).  Only a CSR that passes is dropped; anything else keeps the general layout.  *done says which.
lpp_status pb_from_csr(lpp_engine* e, const DevCsr& A, int64_t n_up, bool* done)
{
	*done = false;
	if (n_up < 512 || A.nrows <= 0 || A.nrows % n_up != 0 || !A.col || !A.val) return LPP_OK;
	const bool cplx = e->is_complex != 0; // complex hoppings: realified in-block matrix, complex couplings (PbState::cplx)
	if (cplx && getenv("LPP_PB_COMPLEX") && atoi(getenv("LPP_PB_COMPLEX")) == 0) return LPP_OK;
	const size_t vd = cplx ? 2 : 1; // doubles per matrix value
	const int64_t n_blk = A.nrows / n_up;
	if (n_blk < 2 || n_blk > 65535) return LPP_OK;
	bool forced = false;
	if (const char* s = getenv("LPP_PRODUCT_LAYOUT")) {
		if (atoi(s) == 0) return LPP_OK;
		forced = true;
	}
	if (!forced && (size_t)A.nrows * vd * sizeof(double) < (cplx ? (size_t)512 << 20 : (size_t)32 << 20)) return LPP_OK; // as for device assembly (assemble_hubbard_pb)
	if (e->cfg.spmv_kernel != LPP_SPMV_AUTO || getenv("LPP_SPMV_KERNEL")) return LPP_OK;
	int want = e->cfg.compress_values;
	if (const char* s = getenv("LPP_COMPRESS_VALUES")) want = atoi(s);
	if (want == 0) return LPP_OK;
	for (const char* k : { "LPP_SHARED_OFFSETS", "LPP_LOCAL16", "LPP_DIAG_CODES", "LPP_BLOCK_TEMPLATE", "LPP_WINDOW_ROWS" })
		if (getenv(k)) return LPP_OK; // switches of the general layout: measure that one
	hipStream_t st = e->stream;
	const int64_t pitch = pb_pitch_for(cplx ? 2 * n_up : n_up); // in doubles
	struct Buf {
		void* p = nullptr;
		~Buf()
		{
			if (p) (void)hipFree(p);
		}
	} d_bad, d_clen, d_cptr, d_ccol, d_cval, d_dval, d_table, d_ov;
	HIP_TRY_MEM(hipMalloc(&d_bad.p, sizeof(int) * 2));
	HIP_TRY(hipMemsetAsync(d_bad.p, 0, sizeof(int) * 2, st));
	// T: block 0's rows, columns inside the block
	std::vector<int64_t> rp0((size_t)n_up + 1);
	HIP_TRY(hipMemcpyAsync(rp0.data(), A.rowptr, sizeof(int64_t) * (size_t)(n_up + 1), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	const int64_t n0 = rp0[(size_t)n_up];
	if (n0 <= 0 || n0 > ((int64_t)1 << 28)) return LPP_OK;
	std::vector<int32_t> c0((size_t)n0);
	std::vector<double> v0((size_t)n0 * vd);
	HIP_TRY(hipMemcpyAsync(c0.data(), A.col, sizeof(int32_t) * (size_t)n0, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(v0.data(), A.val, sizeof(double) * vd * (size_t)n0, hipMemcpyDeviceToHost, st));
	// C: the first row of every block
	HIP_TRY_MEM(hipMalloc(&d_clen.p, sizeof(int64_t) * (size_t)(n_blk + 1)));
	const int nbb = (int)((n_blk + 255) / 256);
	if (cplx)
		k_pb_csr_couplings<false, double2><<<nbb, 256, 0, st>>>(n_up, n_blk, A.rowptr, A.col, (const double2*)A.val, (int64_t*)d_clen.p, nullptr, nullptr, nullptr, (int*)d_bad.p);
	else
		k_pb_csr_couplings<false, double><<<nbb, 256, 0, st>>>(n_up, n_blk, A.rowptr, A.col, (const double*)A.val, (int64_t*)d_clen.p, nullptr, nullptr, nullptr, (int*)d_bad.p);
	std::vector<int64_t> clen((size_t)n_blk), crp((size_t)n_blk + 1, 0);
	HIP_TRY(hipMemcpyAsync(clen.data(), d_clen.p, sizeof(int64_t) * (size_t)n_blk, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipStreamSynchronize(st));
	for (int64_t b = 0; b < n_blk; b++) crp[(size_t)b + 1] = crp[(size_t)b] + clen[(size_t)b];
	const int64_t cz = crp[(size_t)n_blk];
	HIP_TRY_MEM(hipMalloc(&d_cptr.p, sizeof(int64_t) * (size_t)(n_blk + 1)));
	HIP_TRY_MEM(hipMalloc(&d_ccol.p, sizeof(int32_t) * (size_t)std::max<int64_t>(cz, 1)));
	HIP_TRY_MEM(hipMalloc(&d_cval.p, sizeof(double) * vd * (size_t)std::max<int64_t>(cz, 1)));
	HIP_TRY(hipMemcpyAsync(d_cptr.p, crp.data(), sizeof(int64_t) * (size_t)(n_blk + 1), hipMemcpyHostToDevice, st));
	if (cplx)
		k_pb_csr_couplings<true, double2><<<nbb, 256, 0, st>>>(n_up, n_blk, A.rowptr, A.col, (const double2*)A.val, nullptr, (const int64_t*)d_cptr.p, (int32_t*)d_ccol.p,
		                                                      (double2*)d_cval.p, (int*)d_bad.p);
	else
		k_pb_csr_couplings<true, double><<<nbb, 256, 0, st>>>(n_up, n_blk, A.rowptr, A.col, (const double*)A.val, nullptr, (const int64_t*)d_cptr.p, (int32_t*)d_ccol.p,
		                                             (double*)d_cval.p, (int*)d_bad.p);
	std::vector<int32_t> cci((size_t)std::max<int64_t>(cz, 1));
	std::vector<double> cva((size_t)std::max<int64_t>(cz, 1) * vd);
	HIP_TRY(hipMemcpyAsync(cci.data(), d_ccol.p, sizeof(int32_t) * (size_t)cz, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(cva.data(), d_cval.p, sizeof(double) * vd * (size_t)cz, hipMemcpyDeviceToHost, st));
	// D: the stored diagonal of every row, pitched
	const size_t loc = (size_t)n_blk * (size_t)pitch;
	HIP_TRY_MEM(hipMalloc(&d_dval.p, sizeof(double) * loc));
	HIP_TRY(hipMemsetAsync(d_dval.p, 0, sizeof(double) * loc, st));
	const int nbr = (int)std::max<int64_t>(1, std::min<int64_t>((A.nrows + 255) / 256, 1 << 16));
	if (cplx)
		k_pb_csr_diagonal_c<<<nbr, 256, 0, st>>>(n_up, n_blk, pitch, A.rowptr, A.col, (const double2*)A.val, (double*)d_dval.p, (int*)d_bad.p);
	else
		k_pb_csr_diagonal<<<nbr, 256, 0, st>>>(n_up, n_blk, pitch, A.rowptr, A.col, (const double*)A.val, (double*)d_dval.p, (int*)d_bad.p);
	// its distinct values
	HIP_TRY_MEM(hipMalloc(&d_table.p, sizeof(unsigned long long) * kDictTable));
	HIP_TRY_MEM(hipMalloc(&d_ov.p, sizeof(int)));
	HIP_TRY(hipMemsetAsync(d_table.p, 0xff, sizeof(unsigned long long) * kDictTable, st));
	HIP_TRY(hipMemsetAsync(d_ov.p, 0, sizeof(int), st));
	k_dict_collect<<<2048, kBlock, 0, st>>>((const double*)d_dval.p, (int64_t)loc, (unsigned long long*)d_table.p, (int*)d_ov.p);
	std::vector<unsigned long long> host(kDictTable);
	int ov = 0, bad[2] = { 0, 0 };
	HIP_TRY(hipMemcpyAsync(host.data(), d_table.p, sizeof(unsigned long long) * kDictTable, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&ov, d_ov.p, sizeof(int), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(bad, d_bad.p, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	if (bad[0]) return LPP_OK; // a coupling that does not preserve the position, or a row without a stored diagonal
	std::vector<int64_t> trp((size_t)n_up + 1, 0);
	std::vector<int32_t> tci;
	std::vector<double> tva;
	for (int64_t r = 0; r < n_up; r++) {
		for (int64_t p = rp0[(size_t)r]; p < rp0[(size_t)r + 1]; p++)
			if (c0[(size_t)p] >= 0 && c0[(size_t)p] < n_up) { // pb_build skips the diagonal itself
				tci.push_back(c0[(size_t)p]);
				for (size_t k = 0; k < vd; k++) tva.push_back(v0[(size_t)p * vd + k]);
			}
		trp[(size_t)r + 1] = (int64_t)tci.size();
	}
	std::vector<unsigned long long> keys;
	keys.push_back(0ull);
	auto add_key = [&](unsigned long long k) {
		if (std::find(keys.begin(), keys.end(), k) == keys.end()) keys.push_back(k);
	};
	size_t ndiag = 0;
	for (unsigned long long k : host)
		if (k != kDictEmpty) ndiag++;
	bool plain_diag = ov != 0 || ndiag > 250 || (getenv("LPP_PB_PLAIN_DIAG") && atoi(getenv("LPP_PB_PLAIN_DIAG")) != 0);
	if (plain_diag && getenv("LPP_PB_PLAIN_DIAG") && atoi(getenv("LPP_PB_PLAIN_DIAG")) == 0) return LPP_OK;
	if (!plain_diag)
		for (unsigned long long k : host)
			if (k != kDictEmpty) add_key(k);
	for (int64_t p = 0; !cplx && p < cz && keys.size() <= 256; p++) { // (complex couplings: a dictionary of their own, pb_build)
		unsigned long long k;
		std::memcpy(&k, &cva[(size_t)p], 8);
		add_key(k);
	}
	if (keys.size() > 256) return LPP_OK;
	std::sort(keys.begin(), keys.end());
	std::vector<double> dict(256);
	for (size_t i = 0; i < 256; i++) std::memcpy(&dict[i], &keys[std::min(i, keys.size() - 1)], 8);
	// two work vectors + the two parts of a product + the diagonal must fit once the CSR is gone (it is still resident here)
	lpp_status rc;
	if (cplx) {
		std::vector<int64_t> rrp;
		std::vector<int32_t> rci;
		std::vector<double> rva;
		pb_realify(n_up, trp.data(), tci.data(), tva.data(), rrp, rci, rva);
		PbCplxInput cx;
		cx.n_c = n_up;
		cx.t_rp = trp.data();
		cx.t_ci = tci.data();
		cx.t_va = tva.data();
		cx.c_va = cva.data();
		rc = pb_build(e, 2 * n_up, n_blk, rrp.data(), rci.data(), rva.data(), crp.data(), cci.data(), nullptr, dict.data(), (int)keys.size(), 0, -1, 0, 0, &cx);
	} else
		rc = pb_build(e, n_up, n_blk, trp.data(), tci.data(), tva.data(), crp.data(), cci.data(), cva.data(), dict.data(), (int)keys.size());
	if (rc == LPP_ERR_INVALID || rc == LPP_ERR_NOMEM) { // not representable, or no room beside the CSR: the general layout
		if (getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: the product-basis layout does not apply to the uploaded matrix: %s\n", lpp_last_error());
		free_pb(e);
		return LPP_OK;
	}
	if (rc != LPP_OK) return rc;
	PbState& B = e->pb;
	// until every row has been verified the engine must not describe a product-basis matrix: any early return below drops the layout
	struct Undo {
		lpp_engine* e;
		bool* done;
		~Undo()
		{
			if (!*done) free_pb(e);
		}
	} undo { e, done };
	if (B.perm) { // the diagonal was read off the CSR in the basis order: into the stored order (pb.u is free until the first product)
		k_pb_permute<true><<<nbr, 256, 0, st>>>(B.u, (const double*)d_dval.p, B.perm, n_blk, n_up, pitch);
		HIP_TRY(hipMemcpyAsync(d_dval.p, B.u, sizeof(double) * loc, hipMemcpyDeviceToDevice, st));
	}
	if (plain_diag) {
		B.dval = (double*)d_dval.p; // the codes stay 0 (+0.0)
		d_dval.p = nullptr;
	} else {
		k_pb_codes_from_values<<<nbr, 256, 0, st>>>((int64_t)loc, (const double*)d_dval.p, B.dict, B.ndict, B.dcode);
	}
	if (cplx)
		k_pb_csr_verify_c<<<nbr, 256, 0, st>>>(n_up, n_blk, pitch, B.t_ptr, B.t_col, (const double2*)B.t_val, B.c_ptr, B.c_col, B.c_code, B.blockbase, B.dcode, B.dict,
		                                      (const double2*)B.cdict, B.dval, A.rowptr, A.col, (const double2*)A.val, (int*)d_bad.p + 1);
	else
	k_pb_csr_verify<<<nbr, 256, 0, st>>>(n_up, n_blk, pitch, B.t_ptr, B.t_col, B.t_val, B.c_ptr, B.c_col, B.c_code, B.blockbase, B.dcode, B.dict, B.dval, A.rowptr, A.col,
	                                    (const double*)A.val, (int*)d_bad.p + 1, B.inv);
	HIP_TRY(hipMemcpyAsync(bad, d_bad.p, sizeof(int) * 2, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	if (bad[1] || B.nnz != A.nnz) return LPP_OK; // some row is not what (T, C, D) say: not a product-basis matrix (the guard drops the layout)
	*done = true;
	return LPP_OK;
}

// ---- one block: the S = 1/2 Heisenberg chain ------------------------------------------------------------------------------
namespace {
// largest |a - b| and largest |b| as the bit patterns of non-negative doubles (ordered like unsigned integers)
__global__ void k_max_diff(int64_t n, const double* __restrict__ a, const double* __restrict__ b, unsigned long long* __restrict__ out)
{
	double d = 0.0, m = 0.0;
	for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
		d = fmax(d, fabs(a[i] - b[i]));
		m = fmax(m, fabs(b[i]));
		if (a[i] != a[i]) d = 1e300;
	}
	atomicMax(out, (unsigned long long)__double_as_longlong(d));
	atomicMax(out + 1, (unsigned long long)__double_as_longlong(m));
}
__global__ void k_sum_i64(const int64_t* __restrict__ v, int64_t n, unsigned long long* __restrict__ out)
{
	unsigned long long s = 0;
	for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (int64_t)gridDim.x * blockDim.x) s += (unsigned long long)v[k];
	for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
	if ((threadIdx.x & 63) == 0) atomicAdd(out, s);
}
} // namespace

lpp_status pb_chain(lpp_engine* e, const AsmParams& P, int L, int n, const std::vector<double>& hv, bool* done)
{
	*done = false;
	if (e->is_complex || P.nloc <= 0 || P.model != ASM_HEISENBERG) return LPP_OK;
	bool forced = false;
	if (const char* s = getenv("LPP_PRODUCT_LAYOUT")) {
		if (atoi(s) == 0) return LPP_OK;
		forced = true;
	}
	const int64_t n_up = P.nloc, pitch = pb_pitch_for(n_up);
	if (!forced && (size_t)n_up * sizeof(double) < ((size_t)32 << 20)) return LPP_OK; // as for the Hubbard matrices (assemble_hubbard_pb)
	if (e->cfg.spmv_kernel != LPP_SPMV_AUTO || getenv("LPP_SPMV_KERNEL")) return LPP_OK;
	int want = e->cfg.compress_values;
	if (const char* s = getenv("LPP_COMPRESS_VALUES")) want = atoi(s);
	if (want == 0) return LPP_OK;
	for (const char* k : { "LPP_SHARED_OFFSETS", "LPP_LOCAL16", "LPP_DIAG_CODES", "LPP_BLOCK_TEMPLATE", "LPP_WINDOW_ROWS", "LPP_KEEP_PLAIN_CSR" })
		if (getenv(k)) return LPP_OK; // switches of the general layout: measure that one
	if (getenv("LPP_PB_SEG") && atoi(getenv("LPP_PB_SEG")) == 0) return LPP_OK;
	if ((size_t)(pitch + kPbZeroSlots) * sizeof(double) <= (size_t)156 * 1024 && !getenv("LPP_PB_PIECE_ROWS")) return LPP_OK; // a row that fits one LDS window: the general layout's window kernel
	int wcap = 8128;
	if (const char* s = getenv("LPP_PB_PIECE_ROWS")) wcap = (int)std::max<int64_t>(64, std::min<int64_t>(atoll(s), 8128));
	std::vector<int64_t> cnt((size_t)L * L, 0);
	for (size_t k = 0; k < cnt.size(); k++) cnt[k] = hv[k] != 0.0 ? 1 : 0;
	SegPlan SP;
	bool ok = false;
	lpp_status rc = pb_seg_plan_model(L, n, hv, cnt, wcap, SP, &ok, true);
	if (rc != LPP_OK) return rc;
	if (getenv("LPP_VERBOSE"))
		fprintf(stderr, "lpp: chain as one block of the segmented form %s (L = %d, n = %d, %d high sites, %zu segments, %zu items of <= %d positions, <= %d + %d hops per segment)\n",
		        ok ? "planned" : "does not apply", SP.L, SP.n, SP.s, SP.segs.size(), SP.items.size(), SP.wmax, SP.max_cross, SP.max_hh);
	if (!ok || SP.n_up != n_up || SP.nc_pad > 2) return LPP_OK; // (one block per workgroup: the instances with <= 2 pairs of cross hops)
	hipStream_t st = e->stream;
	struct Buf {
		void* p = nullptr;
		~Buf()
		{
			if (p) (void)hipFree(p);
		}
	} d_dval, d_table, d_ov, d_len, d_sum, d_y, d_x, d_ys, d_xs, d_cmp;
	// D: the diagonal of every row straight from the state (the assembler's diag_of: Heisenberg.h:251-275 in the reference's loop order),
	// in the basis order; and the number of entries of the CSR this stands for (the assembler's counting pass)
	const size_t loc = (size_t)pitch;
	HIP_TRY_MEM(hipMalloc(&d_dval.p, sizeof(double) * loc));
	HIP_TRY(hipMemsetAsync(d_dval.p, 0, sizeof(double) * loc, st));
	const int nbr = (int)std::max<int64_t>(1, std::min<int64_t>((n_up + 255) / 256, 1 << 16));
	k_pb_diag_values<ASM_HEISENBERG><<<nbr, kBlock, 0, st>>>(P, pitch, (double*)d_dval.p);
	HIP_TRY_MEM(hipMalloc(&d_len.p, sizeof(int64_t) * (size_t)n_up));
	HIP_TRY_MEM(hipMalloc(&d_sum.p, sizeof(unsigned long long)));
	HIP_TRY(hipMemsetAsync(d_sum.p, 0, sizeof(unsigned long long), st));
	k_asm_count<ASM_HEISENBERG><<<nbr, kBlock, 0, st>>>(P, (int64_t*)d_len.p);
	k_sum_i64<<<1024, 256, 0, st>>>((const int64_t*)d_len.p, n_up, (unsigned long long*)d_sum.p);
	HIP_TRY_MEM(hipMalloc(&d_table.p, sizeof(unsigned long long) * kDictTable));
	HIP_TRY_MEM(hipMalloc(&d_ov.p, sizeof(int)));
	HIP_TRY(hipMemsetAsync(d_table.p, 0xff, sizeof(unsigned long long) * kDictTable, st));
	HIP_TRY(hipMemsetAsync(d_ov.p, 0, sizeof(int), st));
	k_dict_collect<<<2048, kBlock, 0, st>>>((const double*)d_dval.p, (int64_t)loc, (unsigned long long*)d_table.p, (int*)d_ov.p);
	std::vector<unsigned long long> host(kDictTable);
	int ov = 0;
	unsigned long long nnz = 0;
	HIP_TRY(hipMemcpyAsync(host.data(), d_table.p, sizeof(unsigned long long) * kDictTable, hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&ov, d_ov.p, sizeof(int), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipMemcpyAsync(&nnz, d_sum.p, sizeof(nnz), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	(void)hipFree(d_len.p);
	d_len.p = nullptr;
	std::vector<unsigned long long> keys;
	keys.push_back(0ull);
	size_t ndiag = 0;
	for (unsigned long long k : host)
		if (k != kDictEmpty) ndiag++;
	const bool plain_diag = ov != 0 || ndiag > 250;
	if (!plain_diag)
		for (unsigned long long k : host)
			if (k != kDictEmpty && std::find(keys.begin(), keys.end(), k) == keys.end()) keys.push_back(k);
	std::sort(keys.begin(), keys.end());
	std::vector<double> dict(256);
	for (size_t i = 0; i < 256; i++) std::memcpy(&dict[i], &keys[std::min(i, keys.size() - 1)], 8);
	const int64_t c_rp[2] = { 0, 0 };
	rc = pb_build(e, n_up, 1, nullptr, nullptr, nullptr, c_rp, nullptr, nullptr, dict.data(), (int)keys.size(), 0, -1, 0, 0, nullptr, &SP, (int64_t)nnz);
	if (rc == LPP_ERR_INVALID || rc == LPP_ERR_NOMEM) {
		if (getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: the chain keeps the general layout: %s\n", lpp_last_error());
		free_pb(e);
		return LPP_OK;
	}
	if (rc != LPP_OK) return rc;
	PbState& B = e->pb;
	struct Undo { // until the check below has passed the engine must not describe a product-basis matrix
		lpp_engine* e;
		bool* done;
		~Undo()
		{
			if (!*done) free_pb(e);
		}
	} undo { e, done };
	// the diagonal into the stored order of the positions (pb.u is free until the first product)
	k_pb_permute<true><<<nbr, 256, 0, st>>>(B.u, (const double*)d_dval.p, B.perm, 1, n_up, pitch);
	HIP_TRY(hipMemcpyAsync(d_dval.p, B.u, sizeof(double) * loc, hipMemcpyDeviceToDevice, st));
	if (plain_diag) {
		B.dval = (double*)d_dval.p; // the codes stay 0 (+0.0)
		d_dval.p = nullptr;
	} else
		k_pb_codes_from_values<<<nbr, 256, 0, st>>>((int64_t)loc, (const double*)d_dval.p, B.dict, B.ndict, B.dcode);
	// the layout against the model: one product of a random vector through it and through the assembler's row walk (every entry re-derived
	// per row from the term list, in the reference's order), element by element
	HIP_TRY_MEM(hipMalloc(&d_y.p, sizeof(double) * loc));
	HIP_TRY_MEM(hipMalloc(&d_x.p, sizeof(double) * loc));
	HIP_TRY_MEM(hipMalloc(&d_ys.p, sizeof(double) * loc));
	HIP_TRY_MEM(hipMalloc(&d_xs.p, sizeof(double) * loc));
	HIP_TRY_MEM(hipMalloc(&d_cmp.p, sizeof(unsigned long long) * 2));
	HIP_TRY(hipMemsetAsync(d_y.p, 0, sizeof(double) * loc, st));
	HIP_TRY(hipMemsetAsync(d_x.p, 0, sizeof(double) * loc, st));
	HIP_TRY(hipMemsetAsync(d_xs.p, 0, sizeof(double) * loc, st));
	HIP_TRY(hipMemsetAsync(d_cmp.p, 0, sizeof(unsigned long long) * 2, st));
	k_fill_random<<<1024, 256, 0, st>>>((double*)d_y.p, n_up, 0, 4711);
	k_asm_apply<ASM_HEISENBERG, double, false><<<nbr, kBlock, 0, st>>>(P, (const double*)d_y.p, (double*)d_x.p, nullptr, EpiScale { nullptr, nullptr, 0 });
	k_pb_permute<true><<<nbr, 256, 0, st>>>((double*)d_ys.p, (const double*)d_y.p, B.perm, 1, n_up, pitch);
	B.active = true; // (pb_launch reads the state; the guard above drops it again if the check fails)
	e->pitch = pitch;
	pb_launch(e, d_ys.p, d_xs.p, nullptr); // x += H y on a zeroed x
	k_pb_permute<false><<<nbr, 256, 0, st>>>((double*)d_y.p, (const double*)d_xs.p, B.perm, 1, n_up, pitch); // back into the basis order (d_y is free now)
	k_max_diff<<<1024, 256, 0, st>>>(n_up, (const double*)d_y.p, (const double*)d_x.p, (unsigned long long*)d_cmp.p);
	unsigned long long cmp[2] = { 0, 0 };
	HIP_TRY(hipMemcpyAsync(cmp, d_cmp.p, sizeof(cmp), hipMemcpyDeviceToHost, st));
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(st));
	double dmax, xmax;
	std::memcpy(&dmax, &cmp[0], 8);
	std::memcpy(&xmax, &cmp[1], 8);
	if (getenv("LPP_VERBOSE")) fprintf(stderr, "lpp: chain layout against the assembler's row walk: largest difference %.3g of %.3g\n", dmax, xmax);
	if (!(dmax <= 1e-12 * std::max(xmax, 1e-300))) return LPP_OK; // not the same matrix: the general layout (the guard drops this one)
	B.chain_model = true; // (the caller records the model: lpp_engine_get_csr re-runs the assembler from it)
	*done = true;
	return LPP_OK;
}

lpp_status pb_get_csr(lpp_engine* e, int64_t* rowptr, int32_t* colind, void* values)
{
	const PbState& B = e->pb;
	const int64_t n = (B.cplx ? B.n_c : B.n_up) * B.n_blk;
	const size_t vsz = B.cplx ? 2 * sizeof(double) : sizeof(double);
	struct Buf {
		void* p = nullptr;
		~Buf()
		{
			if (p) (void)hipFree(p);
		}
	} drp, dci, dva;
	if (rowptr) HIP_TRY_MEM(hipMalloc(&drp.p, sizeof(int64_t) * (size_t)(n + 1)));
	if (colind || values) {
		HIP_TRY_MEM(hipMalloc(&dci.p, sizeof(int32_t) * (size_t)std::max<int64_t>(B.nnz, 1)));
		HIP_TRY_MEM(hipMalloc(&dva.p, vsz * (size_t)std::max<int64_t>(B.nnz, 1)));
	}
	if (B.cplx)
		k_pb_rebuild_c<<<(int)((n + 255) / 256), 256, 0, e->stream>>>(B.n_c, B.n_blk, B.pitch, B.t_ptr, B.t_col, (const double2*)B.t_val, B.c_ptr, B.c_col, B.c_code,
		                                                             B.blockbase, B.dcode, B.dict, (const double2*)B.cdict, (int64_t*)drp.p, (int32_t*)dci.p, (double2*)dva.p, B.dval);
	else
	k_pb_rebuild<<<(int)((n + 255) / 256), 256, 0, e->stream>>>(B.n_up, B.n_blk, B.pitch, B.t_ptr, B.t_col, B.t_val, B.c_ptr, B.c_col, B.c_code, B.blockbase,
	                                                           B.dcode, B.dict, (int64_t*)drp.p, (int32_t*)dci.p, (double*)dva.p, B.dval, B.inv);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (rowptr) HIP_TRY(hipMemcpy(rowptr, drp.p, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToHost));
	if (colind) HIP_TRY(hipMemcpy(colind, dci.p, sizeof(int32_t) * (size_t)B.nnz, hipMemcpyDeviceToHost));
	if (values) HIP_TRY(hipMemcpy(values, dva.p, vsz * (size_t)B.nnz, hipMemcpyDeviceToHost));
	return LPP_OK;
}

// ---- vector copies that know the pitched layout ------------------------------------------------
lpp_status vec_from_host(lpp_engine* e, double* dev, const void* host)
{
	if (e->tj.active) return tj_vec_from_host(e, dev, host);
	if (e->pitch > 0) {
		const PbState& B = e->pb;
		// stored order of the positions (PbState::perm): the copy lands in pb.u in the basis order and is gathered from there
		double* const land = B.perm ? B.u : dev;
		HIP_TRY(hipMemsetAsync(land, 0, sizeof(double) * (size_t)e->nd_pad, e->stream));
		HIP_TRY(hipMemcpy2DAsync(land, e->esz * (size_t)e->pitch, host, e->esz * (size_t)e->pitch_rows, e->esz * (size_t)e->pitch_rows, (size_t)e->pitch_blocks,
		                         hipMemcpyHostToDevice, e->stream));
		if (B.perm) {
			k_pb_permute<true><<<2048, 256, 0, e->stream>>>(dev, land, B.perm, e->pitch_blocks, e->pitch_rows, e->pitch);
			HIP_TRY(hipGetLastError());
		}
		return LPP_OK;
	}
	HIP_TRY(hipMemcpyAsync(dev, host, e->esz * (size_t)e->n_local, hipMemcpyHostToDevice, e->stream));
	return LPP_OK;
}

lpp_status vec_to_host(lpp_engine* e, void* host, const double* dev)
{
	if (e->tj.active) return tj_vec_to_host(e, host, dev);
	if (e->pitch > 0) {
		const PbState& B = e->pb;
		const double* from = dev;
		if (B.perm) { // back into the basis order, through pb.u (free between products)
			k_pb_permute<false><<<2048, 256, 0, e->stream>>>(B.u, dev, B.perm, e->pitch_blocks, e->pitch_rows, e->pitch);
			HIP_TRY(hipGetLastError());
			from = B.u;
		}
		HIP_TRY(hipMemcpy2DAsync(host, e->esz * (size_t)e->pitch_rows, from, e->esz * (size_t)e->pitch, e->esz * (size_t)e->pitch_rows, (size_t)e->pitch_blocks,
		                         hipMemcpyDeviceToHost, e->stream));
		return LPP_OK;
	}
	HIP_TRY(hipMemcpyAsync(host, dev, e->esz * (size_t)e->n_local, hipMemcpyDeviceToHost, e->stream));
	return LPP_OK;
}

void vec_fill_random(lpp_engine* e, double* dev, uint64_t seed)
{
	if (e->tj.active) {
		tj_fill_random(e, dev, seed);
		return;
	}
	if (e->pitch > 0) {
		if (e->pb.cplx) // complex elements: the stream is indexed by doubles (2 per element), rows and pitch counted in doubles
			k_fill_random_pitched<<<1024, 256, 0, e->stream>>>(dev, e->pitch_blocks, e->pb.n_up, e->pb.pitch, e->row_start * 2, seed, nullptr);
		else
			k_fill_random_pitched<<<1024, 256, 0, e->stream>>>(dev, e->pitch_blocks, e->pitch_rows, e->pitch, e->row_start, seed, e->pb.perm);
		return;
	}
	if (e->nd > 0) k_fill_random<<<1024, 256, 0, e->stream>>>(dev, e->nd, e->row_start * (e->is_complex ? 2 : 1), seed);
}

} // namespace lpp
