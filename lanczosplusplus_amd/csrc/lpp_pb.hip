// lpp_pb.hip -- host side of the product-basis stored layout (kernels and rationale: lpp_pb_kernels.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "lpp_engine_impl.h"

using namespace lpp;

namespace lpp {

int64_t pb_pitch_for(int64_t n_up) { return (n_up + 15) & ~(int64_t)15; }

void free_pb(lpp_engine* e)
{
	PbState& B = e->pb;
	for (void* p : { (void*)B.tw, (void*)B.tw_off, (void*)B.tw_len, (void*)B.t_ptr, (void*)B.t_col, (void*)B.t_val, (void*)B.c_ptr, (void*)B.c_col,
	                 (void*)B.c_code, (void*)B.pace, (void*)B.dict, (void*)B.dcode, (void*)B.blockbase })
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

lpp_status pb_build(lpp_engine* e, int64_t n_up, int64_t n_blk, const int64_t* t_rp, const int32_t* t_ci, const double* t_va,
                    const int64_t* c_rp, const int32_t* c_ci, const double* c_va, const double* dict256, int ndict)
{
	free_pb(e);
	PbState& B = e->pb;
	hipStream_t st = e->stream;
	const int64_t pitch = pb_pitch_for(n_up);
	if ((size_t)(pitch + kPbZeroSlots) * sizeof(double) > (size_t)156 * 1024) return fail(LPP_ERR_INVALID, "pb_build: the block does not fit the LDS window");
	if ((size_t)n_blk * (size_t)pitch * sizeof(double) >= ((size_t)1 << 32)) return fail(LPP_ERR_INVALID, "pb_build: vector beyond 32-bit byte offsets");
	PbTemplate T;
	lpp_status rc = pb_pack_template(n_up, pitch, t_rp, t_ci, t_va, T);
	if (rc != LPP_OK) return rc;
	B.n_up = n_up;
	B.n_blk = n_blk;
	B.pitch = pitch;
	B.G = T.G;
	for (int g = 0; g < kPbGroupsMax; g++) B.gval[g] = T.gval[g];
	B.spb = T.spb;
	B.tw_words = (int64_t)T.words.size();
	B.t_entries = T.entries;
	B.t_slots = T.slots;
	if ((rc = to_device(&B.tw, T.words, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.tw_off, T.off, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.tw_len, T.len, st)) != LPP_OK) return rc;
	// plain off-diagonal copies of T and C (lpp_engine_get_csr walks them; k_pb_down stages C in LDS)
	std::vector<int64_t> tp((size_t)n_up + 1, 0), cp((size_t)n_blk + 1, 0);
	std::vector<int32_t> tc, cc;
	std::vector<double> tv;
	std::vector<uint8_t> ccode;
	for (int64_t r = 0; r < n_up; r++) {
		for (int64_t p = t_rp[r]; p < t_rp[r + 1]; p++) {
			if (t_ci[p] == r) continue;
			if (!tc.empty() && (int64_t)tc.size() > tp[(size_t)r] && tc.back() >= t_ci[p]) return fail(LPP_ERR_INVALID, "pb_build: in-block rows must be sorted by column");
			tc.push_back(t_ci[p]);
			tv.push_back(t_va[p]);
		}
		tp[(size_t)r + 1] = (int64_t)tc.size();
	}
	int64_t longest = 1;
	for (int64_t b = 0; b < n_blk; b++) {
		for (int64_t p = c_rp[b]; p < c_rp[b + 1]; p++) {
			if (c_ci[p] == b) continue;
			if (c_ci[p] < 0 || c_ci[p] >= n_blk) return fail(LPP_ERR_INVALID, "pb_build: block coupling out of range");
			if ((int64_t)cc.size() > cp[(size_t)b] && cc.back() >= c_ci[p]) return fail(LPP_ERR_INVALID, "pb_build: block couplings must be sorted");
			const uint8_t code = code_of(dict256, ndict, c_va[p]);
			if (std::memcmp(&dict256[code], &c_va[p], 8) != 0) return fail(LPP_ERR_INVALID, "pb_build: coupling value missing from the dictionary");
			cc.push_back(c_ci[p]);
			ccode.push_back(code);
		}
		cp[(size_t)b + 1] = (int64_t)cc.size();
		longest = std::max(longest, cp[(size_t)b + 1] - cp[(size_t)b]);
	}
	B.c_nnz = (int64_t)cc.size();
	if ((rc = to_device(&B.t_ptr, tp, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.t_col, tc, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.t_val, tv, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.c_ptr, cp, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.c_col, cc, st)) != LPP_OK) return rc;
	if ((rc = to_device(&B.c_code, ccode, st)) != LPP_OK) return rc;
	std::vector<double> dict(dict256, dict256 + 256);
	if ((rc = to_device(&B.dict, dict, st)) != LPP_OK) return rc;
	B.ndict = ndict;
	// first CSR entry of every block: a block holds Z_T + n_up*(1 + couplings of the block) entries
	std::vector<int64_t> base((size_t)n_blk + 1, 0);
	const int64_t zt = tp[(size_t)n_up];
	for (int64_t b = 0; b < n_blk; b++) base[(size_t)b + 1] = base[(size_t)b] + zt + n_up * (1 + cp[(size_t)b + 1] - cp[(size_t)b]);
	B.nnz = base[(size_t)n_blk];
	if ((rc = to_device(&B.blockbase, base, st)) != LPP_OK) return rc;
	// k_pb_down geometry: one workgroup per CU, 8 groups; the couplings of a workgroup's blocks must fit LDS
	int grid = e->num_cus & ~7;
	if (grid < 8) grid = e->num_cus; // fewer than 8 CUs: one group
	const int slots = grid >= 8 ? grid / 8 : grid;
	B.rowcap = (int)((longest + 7) & ~(int64_t)7);
	B.ids_per_wg = (int)((n_blk + slots - 1) / slots);
	B.down_grid = grid;
	if ((size_t)B.ids_per_wg * (size_t)B.rowcap * 5 > (size_t)150 * 1024) return fail(LPP_ERR_INVALID, "pb_build: block couplings of a workgroup exceed LDS");
	if (!(getenv("LPP_PB_PACE") && atoi(getenv("LPP_PB_PACE")) == 0)) HIP_TRY_MEM(hipMalloc(&B.pace, sizeof(int) * 8 * (size_t)(pitch / 16)));
	HIP_TRY_MEM(hipMalloc(&B.dcode, (size_t)n_blk * (size_t)pitch));
	HIP_TRY(hipMemsetAsync(B.dcode, 0, (size_t)n_blk * (size_t)pitch, st));
	HIP_TRY(hipStreamSynchronize(st));
	e->pitch = pitch;
	e->pitch_rows = n_up;
	e->pitch_blocks = n_blk;
	B.active = true;
	return LPP_OK;
}

// x = beta x + alpha H y (EpiScale semantics of the other product kernels); returns the number of partials written
int pb_launch(lpp_engine* e, const void* y, void* x, double* partial, const EpiScale& sc)
{
	const PbState& B = e->pb;
	hipStream_t st = e->stream;
	// block couplings first (it applies beta), then the in-block part + diagonal, which also forms Re<y|x> of the finished x
	if (B.c_nnz > 0) {
		PbDownArgs d;
		d.pitch = B.pitch;
		d.n_blk = B.n_blk;
		d.npanels = (int)(B.pitch / 16);
		d.ids_per_wg = B.ids_per_wg;
		d.rowcap = B.rowcap;
		d.c_ptr = B.c_ptr;
		d.c_col = B.c_col;
		d.c_code = B.c_code;
		d.dict = B.dict;
		d.y = (const double*)y;
		d.x = (double*)x;
		d.sc = sc;
		d.pace = B.pace;
		d.nwaves = kPbDownThreads / 64;
		if (const char* s = getenv("LPP_PB_DOWN_WAVES")) d.nwaves = std::max(1, std::min(atoi(s), kPbDownThreads / 64));
		if (d.pace) (void)hipMemsetAsync(d.pace, 0, sizeof(int) * 8 * (size_t)d.npanels, st);
		const size_t lds = (size_t)B.ids_per_wg * (size_t)B.rowcap * 5 + 16;
		(void)hipFuncSetAttribute((const void*)k_pb_down, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		k_pb_down<<<B.down_grid, kPbDownThreads, lds, st>>>(d);
	}
	PbUpArgs u;
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
	u.x = (double*)x;
	u.partial = partial;
	u.sc = sc;
	if (B.c_nnz > 0) u.sc.beta_one = 1; // beta was applied by the first kernel
	const size_t lds = sizeof(double) * (size_t)(B.pitch + kPbZeroSlots);
	int nb = (int)std::max<int64_t>(1, std::min<int64_t>(B.n_blk, (int64_t)e->num_cus));
	if (partial) {
		(void)hipFuncSetAttribute((const void*)k_pb_up<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		k_pb_up<true><<<nb, kPbUpThreads, lds, st>>>(u);
	} else {
		(void)hipFuncSetAttribute((const void*)k_pb_up<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		k_pb_up<false><<<nb, kPbUpThreads, lds, st>>>(u);
	}
	return partial ? nb : 0;
}

lpp_status pb_get_csr(lpp_engine* e, int64_t* rowptr, int32_t* colind, void* values)
{
	const PbState& B = e->pb;
	const int64_t n = B.n_up * B.n_blk;
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
		HIP_TRY_MEM(hipMalloc(&dva.p, sizeof(double) * (size_t)std::max<int64_t>(B.nnz, 1)));
	}
	k_pb_rebuild<<<(int)((n + 255) / 256), 256, 0, e->stream>>>(B.n_up, B.n_blk, B.pitch, B.t_ptr, B.t_col, B.t_val, B.c_ptr, B.c_col, B.c_code, B.blockbase,
	                                                           B.dcode, B.dict, (int64_t*)drp.p, (int32_t*)dci.p, (double*)dva.p);
	HIP_TRY(hipGetLastError());
	HIP_TRY(hipStreamSynchronize(e->stream));
	if (rowptr) HIP_TRY(hipMemcpy(rowptr, drp.p, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToHost));
	if (colind) HIP_TRY(hipMemcpy(colind, dci.p, sizeof(int32_t) * (size_t)B.nnz, hipMemcpyDeviceToHost));
	if (values) HIP_TRY(hipMemcpy(values, dva.p, sizeof(double) * (size_t)B.nnz, hipMemcpyDeviceToHost));
	return LPP_OK;
}

// ---- vector copies that know the pitched layout ------------------------------------------------
lpp_status vec_from_host(lpp_engine* e, double* dev, const void* host)
{
	if (e->pitch > 0) {
		HIP_TRY(hipMemsetAsync(dev, 0, sizeof(double) * (size_t)e->nd_pad, e->stream));
		HIP_TRY(hipMemcpy2DAsync(dev, e->esz * (size_t)e->pitch, host, e->esz * (size_t)e->pitch_rows, e->esz * (size_t)e->pitch_rows, (size_t)e->pitch_blocks,
		                         hipMemcpyHostToDevice, e->stream));
		return LPP_OK;
	}
	HIP_TRY(hipMemcpyAsync(dev, host, e->esz * (size_t)e->n_local, hipMemcpyHostToDevice, e->stream));
	return LPP_OK;
}

lpp_status vec_to_host(lpp_engine* e, void* host, const double* dev)
{
	if (e->pitch > 0) {
		HIP_TRY(hipMemcpy2DAsync(host, e->esz * (size_t)e->pitch_rows, dev, e->esz * (size_t)e->pitch, e->esz * (size_t)e->pitch_rows, (size_t)e->pitch_blocks,
		                         hipMemcpyDeviceToHost, e->stream));
		return LPP_OK;
	}
	HIP_TRY(hipMemcpyAsync(host, dev, e->esz * (size_t)e->n_local, hipMemcpyDeviceToHost, e->stream));
	return LPP_OK;
}

void vec_fill_random(lpp_engine* e, double* dev, uint64_t seed)
{
	if (e->pitch > 0) {
		k_fill_random_pitched<<<1024, 256, 0, e->stream>>>(dev, e->pitch_blocks, e->pitch_rows, e->pitch, e->row_start, seed);
		return;
	}
	if (e->nd > 0) k_fill_random<<<1024, 256, 0, e->stream>>>(dev, e->nd, e->row_start * (e->is_complex ? 2 : 1), seed);
}

} // namespace lpp
