// lpp_host.cpp -- host-only parts of the engine (see lpp_host.h).
#include "lpp_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace lpp {

static thread_local std::string g_last_error;

void set_error(const std::string& msg) { g_last_error = msg; }
lpp_status fail(lpp_status code, const std::string& msg)
{
	g_last_error = msg;
	return code;
}

int sturm_count(int n, const double* d, const double* e, double x)
{
	int count = 0;
	double q = 1.0;
	const double tiny = std::numeric_limits<double>::min() * 1e16;
	for (int i = 0; i < n; i++) {
		const double off = (i == 0) ? 0.0 : e[i - 1] * e[i - 1];
		q = d[i] - x - (i == 0 ? 0.0 : off / q);
		if (std::fabs(q) < tiny) q = (q < 0 ? -tiny : tiny);
		if (q < 0) count++;
	}
	return count;
}

double tridiag_kth(int n, const double* d, const double* e, int k)
{
	// Gershgorin interval
	double lo = std::numeric_limits<double>::infinity(), hi = -lo;
	for (int i = 0; i < n; i++) {
		const double r = (i > 0 ? std::fabs(e[i - 1]) : 0.0) + (i + 1 < n ? std::fabs(e[i]) : 0.0);
		lo = std::min(lo, d[i] - r);
		hi = std::max(hi, d[i] + r);
	}
	const double span = std::max(std::fabs(lo), std::fabs(hi));
	lo -= 1e-12 * span + 1e-300;
	hi += 1e-12 * span + 1e-300;
	for (int it = 0; it < 200; it++) {
		const double mid = 0.5 * (lo + hi);
		if (mid <= lo || mid >= hi) break; // interval exhausted in double precision
		if (sturm_count(n, d, e, mid) > k)
			hi = mid;
		else
			lo = mid;
	}
	return 0.5 * (lo + hi);
}

bool tridiag_qr(int n, const double* d_in, const double* e_in, double* w, double* z)
{
	if (n <= 0) return true;
	std::vector<double> d(d_in, d_in + n), e(std::max(n - 1, 1), 0.0);
	for (int i = 0; i + 1 < n; i++) e[i] = e_in[i];
	std::vector<double> Z;
	if (z) {
		Z.assign((size_t)n * n, 0.0);
		for (int i = 0; i < n; i++) Z[(size_t)i * n + i] = 1.0;
	}
	const double eps = std::numeric_limits<double>::epsilon();
	int hi = n - 1;
	int budget = 60 * n + 100;
	while (hi > 0) {
		for (int i = 0; i < hi; i++)
			if (std::fabs(e[i]) <= eps * (std::fabs(d[i]) + std::fabs(d[i + 1]))) e[i] = 0.0;
		if (e[hi - 1] == 0.0) {
			hi--;
			continue;
		}
		int lo = hi - 1;
		while (lo > 0 && e[lo - 1] != 0.0) lo--;
		if (--budget < 0) return false;
		// Wilkinson shift from the trailing 2x2 of the active block
		const double dd = 0.5 * (d[hi - 1] - d[hi]);
		const double ee = e[hi - 1];
		const double denom = dd + (dd >= 0 ? 1.0 : -1.0) * std::hypot(dd, ee);
		const double mu = (denom != 0.0) ? d[hi] - ee * ee / denom : d[hi] - std::fabs(ee);
		double x = d[lo] - mu;
		double bulge = e[lo];
		for (int k = lo; k < hi; k++) {
			const double r = std::hypot(x, bulge);
			const double c = (r == 0.0) ? 1.0 : x / r;
			const double s = (r == 0.0) ? 0.0 : -bulge / r;
			if (k > lo) e[k - 1] = r;
			const double a = d[k], b = e[k], g = d[k + 1];
			d[k] = c * c * a - 2.0 * c * s * b + s * s * g;
			d[k + 1] = s * s * a + 2.0 * c * s * b + c * c * g;
			e[k] = c * s * (a - g) + (c * c - s * s) * b;
			if (k + 1 < hi) {
				bulge = -s * e[k + 1];
				e[k + 1] = c * e[k + 1];
				x = e[k];
			}
			if (z) {
				for (int i = 0; i < n; i++) {
					const double t = Z[(size_t)i * n + k], u = Z[(size_t)i * n + k + 1];
					Z[(size_t)i * n + k] = c * t - s * u;
					Z[(size_t)i * n + k + 1] = s * t + c * u;
				}
			}
		}
	}
	std::vector<int> order(n);
	std::iota(order.begin(), order.end(), 0);
	std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return d[a] < d[b]; });
	for (int k = 0; k < n; k++) w[k] = d[order[k]];
	if (z)
		for (int i = 0; i < n; i++)
			for (int k = 0; k < n; k++) z[(size_t)i * n + k] = Z[(size_t)i * n + order[k]];
	return true;
}

} // namespace lpp

using namespace lpp;

extern "C" {

const char* lpp_last_error(void) { return g_last_error.c_str(); }
int32_t lpp_abi_version(void) { return LPP_ABI_VERSION; }

void lpp_config_default(lpp_config* cfg)
{
	if (!cfg) return;
	std::memset(cfg, 0, sizeof(*cfg));
	cfg->abi_version = LPP_ABI_VERSION;
	cfg->device = 0;
	cfg->dtype = LPP_F64;
	cfg->max_steps = 200;
	cfg->min_steps = 4;
	cfg->reortho = 0;
	cfg->save_vectors = -1;
	cfg->check_lag = 2;
	cfg->spmv_kernel = LPP_SPMV_AUTO;
	cfg->time_kernels = 0;
	cfg->eps = 1e-12;
	cfg->seed = 1234;
	cfg->stream = nullptr;
	cfg->compress_values = -1;
	cfg->reserved = 0;
}

int64_t lpp_xchg_chunk(int64_t n_up, int64_t n_down, int32_t nranks)
{
	if (n_up <= 0 || n_down <= 0 || nranks <= 0) return 0;
	const int64_t per = (n_down + nranks - 1) / nranks, peru = ((n_up + nranks - 1) / nranks + 15) & ~(int64_t)15;
	return per * peru;
}

lpp_status lpp_partition_rows(int64_t nrows, int32_t nranks, int64_t block, int64_t* starts)
{
	if (nrows < 0 || nranks <= 0 || block <= 0 || !starts) return fail(LPP_ERR_INVALID, "lpp_partition_rows: bad argument");
	if (nrows % block != 0) return fail(LPP_ERR_INVALID, "lpp_partition_rows: nrows is not a multiple of block");
	const int64_t nblocks = nrows / block;
	const int64_t per = (nblocks + nranks - 1) / nranks; // ceil: every shard fits the common stride
	for (int32_t r = 0; r <= nranks; r++) starts[r] = std::min<int64_t>((int64_t)r * per, nblocks) * block;
	return LPP_OK;
}

lpp_status lpp_split_csr(int32_t rank, int32_t nranks, const int64_t* shard_starts, int64_t shard_stride,
                         int64_t local_rows, const int64_t* rowptr, const int32_t* colind, const void* values,
                         int32_t elem_bytes, int64_t* nnz_loc, int64_t* nnz_rem, int64_t* rowptr_loc,
                         int32_t* colind_loc, void* values_loc, int64_t* rowptr_rem, int32_t* colind_rem,
                         void* values_rem)
{
	if (rank < 0 || rank >= nranks || !shard_starts || !rowptr || (!colind && rowptr[local_rows] > 0))
		return fail(LPP_ERR_INVALID, "lpp_split_csr: bad argument");
	if (shard_starts[rank + 1] - shard_starts[rank] != local_rows)
		return fail(LPP_ERR_INVALID, "lpp_split_csr: local_rows does not match shard_starts");
	for (int32_t r = 0; r < nranks; r++)
		if (shard_starts[r + 1] - shard_starts[r] > shard_stride)
			return fail(LPP_ERR_INVALID, "lpp_split_csr: shard larger than shard_stride");
	if ((int64_t)nranks * shard_stride > (int64_t)INT32_MAX)
		return fail(LPP_ERR_INVALID, "lpp_split_csr: gathered vector exceeds 32-bit column range");
	const int64_t lo = shard_starts[rank], hi = shard_starts[rank + 1];
	const bool fill = rowptr_loc && rowptr_rem;
	int64_t nl = 0, nr = 0;
	const char* vin = (const char*)values;
	for (int64_t i = 0; i < local_rows; i++) {
		if (fill) {
			rowptr_loc[i] = nl;
			rowptr_rem[i] = nr;
		}
		for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
			const int64_t c = colind[p];
			if (c >= lo && c < hi) {
				if (fill) {
					colind_loc[nl] = (int32_t)(c - lo);
					std::memcpy((char*)values_loc + nl * elem_bytes, vin + p * elem_bytes, elem_bytes);
				}
				nl++;
			} else {
				if (fill) {
					// owner by binary search over shard_starts
					const int64_t* it = std::upper_bound(shard_starts, shard_starts + nranks + 1, c);
					const int64_t owner = (it - shard_starts) - 1;
					if (owner < 0 || owner >= nranks) return fail(LPP_ERR_INVALID, "lpp_split_csr: column out of range");
					colind_rem[nr] = (int32_t)(owner * shard_stride + (c - shard_starts[owner]));
					std::memcpy((char*)values_rem + nr * elem_bytes, vin + p * elem_bytes, elem_bytes);
				}
				nr++;
			}
		}
	}
	if (fill) {
		rowptr_loc[local_rows] = nl;
		rowptr_rem[local_rows] = nr;
	}
	if (nnz_loc) *nnz_loc = nl;
	if (nnz_rem) *nnz_rem = nr;
	return LPP_OK;
}

lpp_status lpp_tridiag_lowest(int32_t n, const double* d, const double* e, int32_t k, double* w, double* z)
{
	if (n <= 0 || k <= 0 || k > n || !d || !w || (n > 1 && !e)) return fail(LPP_ERR_INVALID, "lpp_tridiag_lowest: bad argument");
	if (!z) {
		for (int i = 0; i < k; i++) w[i] = tridiag_kth(n, d, e, i);
		return LPP_OK;
	}
	std::vector<double> ww(n), zz((size_t)n * n);
	if (!tridiag_qr(n, d, e, ww.data(), zz.data())) return fail(LPP_ERR_NOCONV, "lpp_tridiag_lowest: QR iteration did not converge");
	for (int i = 0; i < k; i++) w[i] = ww[i];
	for (int j = 0; j < n; j++)
		for (int i = 0; i < k; i++) z[(size_t)j * k + i] = zz[(size_t)j * n + i];
	return LPP_OK;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------
// product-basis layout: template packing with bank-conflict-free slot assignment
// ---------------------------------------------------------------------------------------------
namespace lpp {

// Proper edge colouring of a bipartite multigraph (left: up to 32 rows, right: nright bank slots) with D = max degree colours
// (Koenig's theorem; alternating-path recolouring).  edges[k] = (row, right vertex); returns the colour of every edge.
int edge_colour(const std::vector<std::pair<int, int>>& edges, int nright, std::vector<int>& colour)
{
	std::vector<int> degL(32, 0), degR((size_t)nright, 0);
	for (const auto& e : edges) {
		degL[(size_t)e.first]++;
		degR[(size_t)e.second]++;
	}
	int D = 0;
	for (int d : degL) D = std::max(D, d);
	for (int d : degR) D = std::max(D, d);
	colour.assign(edges.size(), -1);
	if (D == 0) return 0;
	std::vector<int> atL(32 * (size_t)D, -1), atR((size_t)nright * (size_t)D, -1); // edge holding colour c at a vertex
	for (size_t k = 0; k < edges.size(); k++) {
		const int u = edges[k].first, v = edges[k].second;
		int a = 0, b = 0;
		while (atL[(size_t)u * D + a] >= 0) a++; // free at the row (exists: deg <= D)
		while (atR[(size_t)v * D + b] >= 0) b++; // free at the bank
		if (atR[(size_t)v * D + a] >= 0) {
			// a is taken at v: flip the a/b alternating path that starts at v with colour a.  It cannot end in u
			// (bipartite, a is free at u), afterwards a is free at v.
			std::vector<int> path;
			int x = v, col = a;
			bool right = true;
			for (;;) {
				const int e2 = right ? atR[(size_t)x * D + col] : atL[(size_t)x * D + col];
				if (e2 < 0) break;
				path.push_back(e2);
				x = right ? edges[(size_t)e2].first : edges[(size_t)e2].second;
				right = !right;
				col = (col == a) ? b : a;
			}
			for (int e2 : path) {
				atL[(size_t)edges[(size_t)e2].first * D + colour[(size_t)e2]] = -1;
				atR[(size_t)edges[(size_t)e2].second * D + colour[(size_t)e2]] = -1;
			}
			for (int e2 : path) {
				const int nc = (colour[(size_t)e2] == a) ? b : a;
				colour[(size_t)e2] = nc;
				atL[(size_t)edges[(size_t)e2].first * D + nc] = e2;
				atR[(size_t)edges[(size_t)e2].second * D + nc] = e2;
			}
		}
		colour[k] = a;
		atL[(size_t)u * D + a] = (int)k;
		atR[(size_t)v * D + a] = (int)k;
	}
	return D;
}

lpp_status pb_pack_template(int64_t rows, int64_t pitch, const int64_t* rp, const int32_t* ci, const double* va, PbTemplate& out, int bank_ways, int64_t window)
{
	if (rows <= 0 || pitch < rows || (pitch & 15) != 0) return fail(LPP_ERR_INVALID, "pb_pack_template: bad shape");
	if (window == 0 && pitch + kPbZeroSlotsHost > 65536) return fail(LPP_ERR_INVALID, "pb_pack_template: bad shape");
	if (window != 0 && (window < 64 || (window & 63) != 0 || window + kPbZeroSlotsHost > 65536 || rows >= ((int64_t)1 << 24)))
		return fail(LPP_ERR_INVALID, "pb_pack_template: window must be a multiple of 64 below 65504, rows below 2^24");
	if (bank_ways < 1 || bank_ways > 4) return fail(LPP_ERR_INVALID, "pb_pack_template: bank_ways must be 1..4");
	out = PbTemplate();
	out.W = window;
	const int64_t zero_at = window ? window : pitch; // window index of the first zero slot
	// value groups (bit patterns, first-seen order then sorted for determinism)
	std::vector<uint64_t> keys;
	for (int64_t r = 0; r < rows; r++)
		for (int64_t p = rp[r]; p < rp[r + 1]; p++) {
			if (ci[p] == r) continue;
			if (ci[p] < 0 || ci[p] >= rows) return fail(LPP_ERR_INVALID, "pb_pack_template: column out of range");
			uint64_t k;
			std::memcpy(&k, &va[p], 8);
			if (std::find(keys.begin(), keys.end(), k) == keys.end()) {
				keys.push_back(k);
				if ((int)keys.size() > kPbGroupsMax) return fail(LPP_ERR_INVALID, "pb_pack_template: more than 8 distinct in-block values");
			}
		}
	std::sort(keys.begin(), keys.end());
	out.G = std::max<int>(1, (int)keys.size());
	for (size_t g = 0; g < keys.size(); g++) std::memcpy(&out.gval[g], &keys[g], 8);
	const int G = out.G;
	out.spb = (int)((rows + 63) / 64);
	out.off.assign((size_t)out.spb * G, 0);
	out.len.assign((size_t)out.spb * G, 0);
	if (window) {
		out.foff.assign((size_t)out.spb, 0);
		out.flen.assign((size_t)out.spb, 0);
	}
	std::vector<std::pair<int, int>> edges;
	std::vector<int> ecol, colour;
	std::vector<int> slot_idx; // [slot][lane] window index or -1
	for (int j = 0; j < out.spb; j++) {
		// the piece this slice belongs to: columns [c0, c1) are read from the LDS window, the others from memory
		const int64_t c0 = window ? ((int64_t)j * 64 / window) * window : 0, c1 = window ? std::min<int64_t>(c0 + window, rows) : rows;
		if (window) {
			struct Far {
				int lane, g;
				int64_t col;
			};
			std::vector<Far> far;
			for (int l = 0; l < 64; l++) {
				const int64_t r = (int64_t)j * 64 + l;
				if (r >= rows) break;
				for (int64_t p = rp[r]; p < rp[r + 1]; p++) {
					if (ci[p] == r || (ci[p] >= c0 && ci[p] < c1)) continue;
					uint64_t k;
					std::memcpy(&k, &va[p], 8);
					const int g = (int)(std::find(keys.begin(), keys.end(), k) - keys.begin());
					far.push_back(Far { l, g, (int64_t)ci[p] });
				}
			}
			// groups of equal column offset, largest first; a group takes the first slot in which all its lanes are free
			std::vector<size_t> order(far.size());
			for (size_t k = 0; k < far.size(); k++) order[k] = k;
			auto offs = [&](size_t k) { return far[k].col - ((int64_t)j * 64 + far[k].lane); };
			std::sort(order.begin(), order.end(), [&](size_t x, size_t y) { return offs(x) != offs(y) ? offs(x) < offs(y) : far[x].lane < far[y].lane; });
			struct Grp {
				size_t first, count;
			};
			std::vector<Grp> groups;
			for (size_t k = 0; k < order.size();) {
				size_t k2 = k;
				while (k2 < order.size() && offs(order[k2]) == offs(order[k])) k2++;
				groups.push_back(Grp { k, k2 - k });
				k = k2;
			}
			std::stable_sort(groups.begin(), groups.end(), [](const Grp& x, const Grp& y) { return x.count > y.count; });
			std::vector<uint64_t> busy; // lanes taken, per slot
			std::vector<uint32_t> fw; // [slot][lane]
			for (const Grp& gr : groups) {
				// a row may hold two entries of one offset only if it holds the same column twice: never in a CSR row, so a group has one entry per lane
				uint64_t mask = 0;
				for (size_t k = 0; k < gr.count; k++) mask |= 1ull << far[order[gr.first + k]].lane;
				size_t s = 0;
				while (s < busy.size() && (busy[s] & mask)) s++;
				if (s == busy.size()) {
					busy.push_back(0);
					fw.resize(fw.size() + 64, 0u);
				}
				busy[s] |= mask;
				for (size_t k = 0; k < gr.count; k++) {
					const Far& f = far[order[gr.first + k]];
					fw[s * 64 + (size_t)f.lane] = (uint32_t)f.col | ((uint32_t)f.g << 24) | 0x80000000u;
				}
			}
			while (busy.size() & 3) { // whole groups of 4 slots (the kernel's unit), the rest is filling
				busy.push_back(0);
				fw.resize(fw.size() + 64, 0u);
			}
			if (busy.size() > 65535) return fail(LPP_ERR_INVALID, "pb_pack_template: too many entries leave the window");
			for (size_t s = 0; s < busy.size(); s++)
				for (int l = 0; l < 64; l++) {
					uint32_t& w = fw[s * 64 + (size_t)l];
					if (w & 0x80000000u)
						w &= 0x7fffffffu;
					else // filling: the row's own element (a valid address inside the vector) times 0.0
						w = (uint32_t)std::min<int64_t>((int64_t)j * 64 + l, rows - 1) | ((uint32_t)G << 24);
				}
			out.foff[(size_t)j] = (int32_t)(out.fwords.size() / 64);
			out.flen[(size_t)j] = (uint16_t)busy.size();
			out.fwords.insert(out.fwords.end(), fw.begin(), fw.end());
			out.far_entries += (int64_t)far.size();
			if (out.fwords.size() > ((size_t)1 << 30)) return fail(LPP_ERR_INVALID, "pb_pack_template: template too large");
		}
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
					const int64_t r = (int64_t)j * 64 + h * 32 + l;
					if (r >= rows) break;
					for (int64_t p = rp[r]; p < rp[r + 1]; p++) {
						if (ci[p] == r) continue;
						if (ci[p] < c0 || ci[p] >= c1) continue; // leaves the window: kept in the far list above
						uint64_t k;
						std::memcpy(&k, &va[p], 8);
						if (!keys.empty() && k != keys[(size_t)g]) continue;
						// a bank serves `bank_ways` different addresses per slot: its entries are dealt over that many right vertices
						const int bank = (int)(ci[p] & 31); // c0 is a multiple of 64: the bank of the window index
						edges.emplace_back(l, bank * bank_ways + (seen[bank]++ % bank_ways));
						ecol.push_back((int)(ci[p] - c0));
					}
				}
				const int D = edge_colour(edges, 32 * bank_ways, colour);
				nslots = std::max(nslots, D);
				for (size_t k = 0; k < edges.size(); k++) ents.push_back(Ent { h * 32 + edges[k].first, ecol[k], colour[k] });
				out.entries += (int64_t)edges.size();
			}
			// whole chunks of 4 slots (one 16-byte load per lane); the filling reads zero slots
			const int nchunks = (nslots + 3) / 4;
			nslots = nchunks * 4;
			slot_idx.assign((size_t)nslots * 64, -1);
			for (const Ent& e : ents) slot_idx[(size_t)e.slot * 64 + e.lane] = e.col;
			// padding: a zero slot in the bank that the real entries of this slot and half-wave use least
			for (int s = 0; s < nslots; s++)
				for (int h = 0; h < 2; h++) {
					int used[32] = { 0 };
					bool any_pad = false;
					for (int l = 0; l < 32; l++) {
						const int c = slot_idx[(size_t)s * 64 + h * 32 + l];
						if (c >= 0)
							used[c & 31]++;
						else
							any_pad = true;
					}
					if (!any_pad) continue;
					int z = 0;
					for (int t = 1; t < 32; t++)
						if (used[(zero_at + t) & 31] < used[(zero_at + z) & 31]) z = t;
					for (int l = 0; l < 32; l++)
						if (slot_idx[(size_t)s * 64 + h * 32 + l] < 0) slot_idx[(size_t)s * 64 + h * 32 + l] = (int)(zero_at + z);
				}
			out.off[(size_t)j * G + g] = (int32_t)(out.words.size() / 128); // in chunks of 64 lanes x 2 words
			out.len[(size_t)j * G + g] = (uint16_t)nchunks;
			for (int c = 0; c < nchunks; c++)
				for (int l = 0; l < 64; l++)
					for (int k = 0; k < 4; k += 2) // two 16-bit window indices per word: slot 4c+k in the low half, 4c+k+1 in the high half
						out.words.push_back((uint32_t)slot_idx[(size_t)(4 * c + k) * 64 + l] | ((uint32_t)slot_idx[(size_t)(4 * c + k + 1) * 64 + l] << 16));
			out.slots += (int64_t)nslots * 64;
			if (out.words.size() > ((size_t)1 << 30)) return fail(LPP_ERR_INVALID, "pb_pack_template: template too large");
		}
	}
	out.words.resize(out.words.size() + 128 * 8, (uint32_t)zero_at | ((uint32_t)zero_at << 16)); // slack for the look-ahead loads
	if (window) out.fwords.resize(out.fwords.size() + 64 * 8, (uint32_t)G << 24); // the same for the far lists (row 0 times 0.0)
	return LPP_OK;
}

} // namespace lpp

extern "C" lpp_status lpp_pb_pack_template(int64_t rows, int64_t pitch, const int64_t* rowptr, const int32_t* colind, const double* values,
                                           int32_t* ngroups, double* group_values, int32_t* slices, int64_t* nwords, int32_t* off, uint16_t* len,
                                           uint32_t* words, int64_t* entries, int64_t* slots, int32_t bank_ways)
{
	if (!rowptr || !ngroups || !nwords) return lpp::fail(LPP_ERR_INVALID, "lpp_pb_pack_template: null argument");
	lpp::PbTemplate t;
	lpp_status st = lpp::pb_pack_template(rows, pitch, rowptr, colind, values, t, bank_ways);
	if (st != LPP_OK) return st;
	*ngroups = t.G;
	*nwords = (int64_t)t.words.size();
	if (slices) *slices = t.spb;
	if (entries) *entries = t.entries;
	if (slots) *slots = t.slots;
	if (group_values) std::memcpy(group_values, t.gval, sizeof(double) * lpp::kPbGroupsMax);
	if (off) std::memcpy(off, t.off.data(), sizeof(int32_t) * t.off.size());
	if (len) std::memcpy(len, t.len.data(), sizeof(uint16_t) * t.len.size());
	if (words) std::memcpy(words, t.words.data(), sizeof(uint32_t) * t.words.size());
	return LPP_OK;
}
