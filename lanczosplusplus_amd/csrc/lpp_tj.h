// lpp_tj.h -- the description of the one-orbital t-J model in its hole-major form (rationale: lpp_tj_kernels.h): plain structures shared by
// the host planner (lpp_tj_host.cpp, no device code), the kernel (lpp_tj_kernels.h) and the engine (lpp_tj.hip).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace lpp {

constexpr int kTjMaxPairs = 96; // bonds with both sites occupied, per hole configuration
constexpr int kTjMaxHops = 64; // (electron, neighbouring hole) moves per hole configuration
constexpr int kTjThreads = 256;
#ifndef LPP_TJ_ROWS
#define LPP_TJ_ROWS 2
#endif
#ifndef LPP_TJ_WINDOW
#define LPP_TJ_WINDOW 1024
#endif
constexpr int kTjRowsPerThread = LPP_TJ_ROWS; // rows of a thread per round (their gathers are in flight together)
constexpr int kTjWindow = LPP_TJ_WINDOW; // patterns of one work item at most: its run of the vector is staged in LDS
constexpr int kTjMaxHalf = 12; // bits of a half pattern (rank tables in LDS: 4096 x (4 + 2) bytes at most)

struct TjPair { // 16 bytes
	uint32_t mask; // bit p | bit q (compressed positions)
	uint32_t pad;
	double v; // 0.5 J(i,j) (-1)^(q - p)
};
struct TjHop { // 24 bytes
	int32_t dst; // block of the bra
	uint8_t lo, m, dir, pad; // bits [lo, lo + m] of sigma rotate: dir 0 the electron sits at lo and moves to lo + m, dir 1 it sits at lo + m and moves to lo
	double vr, vi;
};
struct TjBlock { // 16 bytes
	int32_t x_first, h_first;
	int16_t nx, nxl; // bonds; of these the first nxl have both positions among the low kbits of the pattern
	int16_t nh, pad;
};
struct TjItem { // a run of whole segments (patterns sharing the bits above the low kbits): a flip among the low bits stays inside
	int32_t r0, len;
};

// host copy of what lpp_engine_assemble_tj / lpp_engine_set_model_tj was given: lpp_engine_get_csr re-runs the device assembler from it
struct TjModel {
	int L = 0, nup = 0, ndown = 0, npot = 0;
	bool has_im = false, has_pv = false;
	std::vector<double> hop_re, hop_im, jpm, jzz, w, pv;
};

// the plan: everything the kernel reads except the diagonal and the boundary's permutation (which come from the device assembler)
struct TjPlan {
	int Lo = 0, lb = 0, hb = 0, ns = 0, nblk = 0, kbits = 0;
	bool cplx_hops = false;
	std::vector<uint32_t> pat; // spin patterns of the occupied sites, ascending
	std::vector<uint32_t> holes; // hole words, ascending: block b = holes[b]
	std::vector<int32_t> hi_base; // rank(s) = hi_base[s >> lb] + lo_rank[s & ((1 << lb) - 1)]
	std::vector<uint16_t> lo_rank;
	std::vector<TjItem> items;
	std::vector<TjBlock> blocks;
	std::vector<TjPair> pairs;
	std::vector<TjHop> hops;
};
// *ok == false: the form does not apply (why says so); sizes are the caller's to check (tj_applies)
void tj_plan(const TjModel& M, TjPlan& P, bool* ok, std::string* why);

} // namespace lpp
