// lpp_host.h -- host-only pieces of the engine (no HIP): error plumbing, the symmetric
// tridiagonal eigen-solver used by the convergence test / Ritz reconstruction, the 1-D row
// partition and the local/remote CSR split.  Compiled into liblpp_engine.so; also exercised
// by the CPU test-suite (no GPU needed).
#pragma once
#include <stdint.h>
#include <string>
#include <utility>
#include <vector>

#include "../../include/lpp_engine.h"

namespace lpp {

void set_error(const std::string& msg);
lpp_status fail(lpp_status code, const std::string& msg);

// number of eigenvalues of tridiag(d,e) strictly below x (Sturm count)
int sturm_count(int n, const double* d, const double* e, double x);
// k-th (0-based) eigenvalue by bisection
double tridiag_kth(int n, const double* d, const double* e, int k);
// full decomposition by the implicit symmetric QR step with Wilkinson shift (Givens bulge
// chasing, deflation from the bottom); eigenvalues ascending in w, eigenvectors in the columns
// of z (row-major n x n).  Returns false when it does not converge.
bool tridiag_qr(int n, const double* d, const double* e, double* w, double* z);

} // namespace lpp

// ---------------------------------------------------------------------------------------------
// Product-basis layout (lpp_pb_kernels.h), host part: the in-block matrix T as per-slice, per-value-group streams of
// LDS window addresses.  Within a half-wave (32 lanes = 32 consecutive rows) the entries of a group are assigned to
// slots by a bipartite edge colouring (rows x LDS banks) so that no bank is asked for more than `bank_ways` different
// addresses in a slot: bank_ways = 1 is conflict-free (a ds_read_b64 gather in its 2 cycles instead of ~3.7x that for
// entries in column order) at the price of more slots (filled with reads of zero slots); 2 halves the worst bank's share.
// ---------------------------------------------------------------------------------------------
namespace lpp {

constexpr int kPbGroupsMax = 8;
constexpr int kPbZeroSlotsHost = 32;

struct PbTemplate {
	int G = 0; // value groups
	double gval[kPbGroupsMax] = { 0 };
	int spb = 0; // 64-row slices
	std::vector<int32_t> off; // [spb*G] first chunk of (slice, group)
	std::vector<uint16_t> len; // [spb*G] chunks (a chunk = 4 slots)
	std::vector<uint32_t> words; // chunk c, lane l at [(c*64 + l)*2 + {0,1}]: window indices of slots 4c, 4c+1 | 4c+2, 4c+3 (16 bits each, low half first)
	int64_t entries = 0; // real entries
	int64_t slots = 0; // lane-slots incl. padding
	// rows beyond one LDS window (pb_pack_template with window W > 0): the row is cut into pieces of W consecutive positions
	// (W a multiple of 64, so a slice never straddles two pieces); the streams above hold the entries whose column lies in the
	// row's OWN piece, as indices relative to the piece (zero slots at index W), and the entries that leave the piece are kept
	// per slice as 32-bit words  column (24 bits) | value group << 24  (group G = filling, multiplies by 0.0, column = the row
	// itself), slot-major.  Entries with the same column offset share a slot wherever their lanes are free: basis words are
	// ascending, so a hop among the high sites moves 64 consecutive rows by the same amount and the slot is ONE coalesced read.
	int64_t W = 0;
	std::vector<int32_t> foff; // [spb] first far slot of the slice (a slot = 64 words)
	std::vector<uint16_t> flen; // [spb] far slots
	std::vector<uint32_t> fwords;
	int64_t far_entries = 0;
};

// rows x rows CSR (rp, ci, va); entries with ci == row (the diagonal) are skipped.  pitch: the zero slots start at window
// index `pitch` (a multiple of 16, >= rows; pitch + 32 <= 65536).  Fails (LPP_ERR_INVALID) when the off-diagonal part has more
// than kPbGroupsMax distinct values.
// window > 0: pieces of `window` positions (see PbTemplate::W); the zero slots then sit at window index `window`.
lpp_status pb_pack_template(int64_t rows, int64_t pitch, const int64_t* rp, const int32_t* ci, const double* va, PbTemplate& out, int bank_ways = 2,
                            int64_t window = 0);

// Proper edge colouring of a bipartite multigraph (left: up to 32 rows, right: nright bank slots) with D = max degree colours;
// edges[k] = (row, right vertex); returns D and the colour of every edge.
int edge_colour(const std::vector<std::pair<int, int>>& edges, int nright, std::vector<int>& colour);

} // namespace lpp

// ---------------------------------------------------------------------------------------------
// Segmented in-block form for rows beyond one LDS window (lpp_pbseg.h): the host-side plan.
// ---------------------------------------------------------------------------------------------
#include "lpp_pbseg.h"
namespace lpp {

struct SegPlan {
	int L = 0, n = 0, s = 0; // sites, particles of the species; high sites
	int64_t n_up = 0;
	int G = 1; // value groups of the low-low hops (1 or 2)
	double gval[kPbGroupsMax] = { 0 };
	std::vector<int32_t> perm, inv; // stored position -> basis index and back
	std::vector<SegInst> segs;
	std::vector<SegCross> cross;
	std::vector<SegHh> hh;
	std::vector<SegItem> items;
	std::vector<SegSlice> slices;
	std::vector<uint32_t> words; // in-window lists, as PbTemplate::words
	std::vector<uint32_t> xwords; // cross tables: two 16-bit words (two hops of one high site) per position
	int wmax = 0, zmax = 0, ws = 0, ntypes = 0; // longest item, largest zero_at, window stride (elements), item types
	int max_cross = 0, max_hh = 0; // most cross / high-high hops of a segment
	int nc_pad = 0, nh_pad = 0; // entries per segment in `cross` / `hh` (padded with value 0.0): the kernel instance's NC / NH
	int pre0 = 4, pre1 = 4; // list chunks of value group 0 / 1 the kernel requests ahead (k_pb_up_seg's P0 / P1)
	int64_t entries = 0, entries_lo = 0, slots_lo = 0; // rows verified; low-low entries / lane-slots of the item types
};

// T: n_up x n_up CSR (diagonal entries skipped).  *ok = false: T is not the hopping matrix of one species in the ascending-word basis,
// or something exceeds the kernel's limits -- the caller keeps the per-position template.  wcap: longest segment / item (<= 8128).
lpp_status pb_seg_plan(int64_t n_up, const int64_t* rp, const int32_t* ci, const double* va, int wcap, SegPlan& out, bool* ok);
// the same from the species itself: L sites, n particles, amplitudes hv[to * L + from] (entry = hv x (-1)^(particles strictly between the two
// sites)), cnt[to * L + from] != 0 where a hop exists.  Nothing to verify against: the caller checks the layout it builds (pb_chain).
lpp_status pb_seg_plan_model(int L, int n, const std::vector<double>& hv, const std::vector<int64_t>& cnt, int wcap, SegPlan& out, bool* ok, bool one_block = false);

} // namespace lpp
