// lpp_pbseg.h -- the in-block matrix T of a product-basis Hamiltonian (lpp_pb_kernels.h) for one-species spaces BEYOND one LDS
// window, decomposed by the HIGH sites of the species' basis word (round 4; replaces the per-position template of k_pb_up_big2 for
// BASELINE config 5's sectors, which streamed 80 bytes per position -- 6.2 MB per row at N_up = 77520 -- through every XCD's L2).
//
// T is the hopping matrix of one species (HubbardHelper.h:191-243) in the basis BasisOneSpin.h:53-61 builds: all L-bit words of n set
// bits, ascending.  Ascending order makes the top s sites the major sort key, so a block's row is a sequence of SEGMENTS, one per
// configuration t of the top s sites, segment t holding the C(L-s, n-|t|) configurations of the low L-s sites in ascending order.
// A hop then is one of three things:
//   low-low    both sites low: stays inside the segment, and the sub-matrix is THE SAME for every segment with the same number of
//              low particles (it does not see t).  Kept as 16-bit LDS window indices exactly as in k_pb_up, once per item TYPE
//              (an item = consecutive segments that share one LDS window); ~0.8 MB for the whole 4x5 lattice instead of 6.2 MB per row.
//   high-high  both sites high: maps segment t onto segment t' position by position with ONE sign (the particles between the two
//              sites are all high).  No words at all: a (source segment, value) pair per segment, read as coalesced runs.
//   cross      one low site i, one high site j: source = (t ^ bit j, lo ^ bit i).  The map lo -> rank(lo ^ bit i) depends only on
//              (number of low particles, i, add/remove), the sign factorises into (low bits above i) x (high bits below j): a 16-bit
//              word per position in a table shared by all segments of a class, and a (table, source segment, value) triple per segment.
//              The hops of one high site share direction and source segment: their words travel in pairs (one 32-bit load for two hops).
// Everything position-dependent is shared by class; everything segment-dependent is a few scalars.  The whole description of T is
// ~1.5 MB for the (7) species of the 4x5 lattice and stays in every XCD's L2 next to the rows in flight.
//
// The stored order of the positions inside a block is the segments sorted by length (PbState::perm: only the boundary knows), so
// that an item is one contiguous run of stored positions.
#pragma once
#include <stdint.h>

namespace lpp {

constexpr int kSegMaxCross = 6; // PAIRS of cross hops per segment the kernel carries (every segment's list is padded to the instance's NC with value 0.0)
constexpr int kSegMaxHh = 16; // high-high entries per segment (one block per workgroup -- chains up to 17 high sites; two blocks per workgroup: 12)
__host__ __device__ constexpr int pb_seg_hh_cap(int rows) { return rows == 1 ? kSegMaxHh : 12; } // entries per segment in the LDS copy
constexpr int kSegMaxSegs = 16; // segments per item (LDS tables)
constexpr int kSegWinPad = 2; // window index of an item's first position (the staged run starts at an even element)

struct SegCross { // 32 bytes: the (up to) two cross hops of one high site that share a direction -- same source segment, one 32-bit word per position
	int32_t wordoff; // first word of the (class, low site a, low site b, add/remove) table in xwords
	int32_t srcbase; // stored position of the source segment's first element
	double val[2]; // hopping value x sign of the high bits below the high site, for the low / high half of the word (0.0: no second hop)
	int64_t pad;
};
struct SegHh { // 16 bytes
	int32_t srcbase;
	int32_t pad; // 1: a hop (lane l reads srcbase + its offset in the segment); 0: filling entry of value 0.0, every lane reads element srcbase
	double val;
};
struct SegInst { // 32 bytes, stored order
	int32_t sbase, len;
	int32_t cross_first, ncross;
	int32_t hh_first, nhh;
	int32_t pad0, pad1;
};
struct SegItem { // 32 bytes
	int32_t c0, wlen; // stored positions [c0, c0 + wlen)
	int32_t seg_first, nseg;
	int32_t slice_first, nslices; // the item type's slices
	int32_t zero_at; // window index of the 32 zero slots (filling entries)
	int32_t type;
};
struct SegSlice { // 16 bytes: 64 consecutive positions of ONE segment and the heads of its low-low lists
	uint16_t first; // offset in the item
	uint8_t count; // valid lanes (1..64)
	uint8_t seg; // segment of the item
	uint16_t segoff; // offset in the segment (a multiple of 64)
	uint8_t nc[2]; // chunks of value group 0 / 1
	int32_t off[2]; // their first chunk
};
constexpr int kSegMaxSlices = 160; // slices per item (LDS copy of the heads)

// a cross word (16 bits, two per 32-bit table entry): bits 0..12 offset in the source segment, bits 14..15 = +1 (01), -1 (11) or no entry (00)
// as a signed 2-bit field
constexpr uint16_t kSegWordPlus = 0x4000, kSegWordMinus = 0xC000;

} // namespace lpp
