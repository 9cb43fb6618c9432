// Bases.h -- host many-body bases with the reference's class names and indexing (bit-exact):
//   BasisOneSpin / BasisHubbardLanczos   src/Models/HubbardOneOrbital/{BasisOneSpin.h,BasisHubbardLanczos.h}
//   BasisHeisenberg (S = 1/2)            src/Models/Heisenberg/BasisHeisenberg.h
//   BasisTjMultiOrbLanczos (orbitals=1)  src/Models/TjMultiOrb/BasisTjMultiOrbLanczos.h
// perfectIndex is O(L) here (combinatorial ranking) where the reference scans linearly
// (BasisHeisenberg.h:73-80) or bisects a stored list (BasisTjMultiOrbLanczos.h:70-107); the bases are
// sorted ascending, so the returned index is the same.
#ifndef LPP_HOST_BASES_H
#define LPP_HOST_BASES_H

#include "Compat.h"

namespace LanczosPlusPlus {

using LppHost::err;

struct ProgramGlobals {
	typedef unsigned long int WordType;
	enum { SPIN_UP, SPIN_DOWN };
	static WordType bitmask(SizeType i) { return WordType(1) << i; }
	// parity of the set bits of a strictly below position i (ProgramGlobals.h:109-114)
	static int doSign(WordType a, SizeType i)
	{
		const WordType mask = (WordType(1) << i) - 1;
		return (__builtin_popcountl(a & mask) & 1) ? -1 : 1;
	}
};

class Binomial {
public:
	static const Binomial& get()
	{
		static Binomial b;
		return b;
	}
	SizeType operator()(SizeType n, SizeType m) const { return (n < N && m < N) ? c_[n][m] : 0; }

private:
	enum { N = 66 };
	Binomial()
	{
		for (SizeType n = 0; n < N; n++)
			for (SizeType m = 0; m < N; m++) c_[n][m] = 0;
		for (SizeType n = 0; n < N; n++) {
			c_[n][0] = 1;
			for (SizeType m = 1; m <= n; m++) {
				const unsigned __int128 v = (unsigned __int128)c_[n - 1][m - 1] + c_[n - 1][m];
				c_[n][m] = v > (unsigned __int128)~SizeType(0) ? ~SizeType(0) : (SizeType)v;
			}
		}
	}
	SizeType c_[N][N];
};

// all nsite-bit words with npart set bits, ascending (BasisOneSpin.h:46-61)
class BasisOneSpin {
public:
	typedef ProgramGlobals::WordType WordType;
	BasisOneSpin(SizeType nsite, SizeType npart) : nsite_(nsite), npart_(npart)
	{
		if (npart > nsite) err("BasisOneSpin: more particles than sites\n");
		const SizeType hilbert = Binomial::get()(nsite, npart);
		data_.reserve(hilbert);
		if (npart == 0) {
			data_.push_back(0);
			return;
		}
		WordType w = (WordType(1) << npart) - 1;
		for (SizeType i = 0; i < hilbert; i++) {
			data_.push_back(w);
			// next word with the same popcount (Gosper)
			const WordType c = w & (~w + 1), r = w + c;
			w = r ? (((r ^ w) >> 2) / c) | r : 0;
		}
	}
	SizeType size() const { return data_.size(); }
	const WordType& operator[](SizeType i) const { return data_[i]; }
	SizeType electrons() const { return npart_; }
	// sum over set bits b (c-th from below) of C(b,c)  (BasisOneSpin.h:73-81)
	static SizeType perfectIndex(WordType state)
	{
		const Binomial& comb = Binomial::get();
		SizeType n = 0;
		for (SizeType c = 1; state; c++) {
			const SizeType b = __builtin_ctzl(state);
			n += comb(b, c);
			state &= state - 1;
		}
		return n;
	}

private:
	SizeType nsite_, npart_;
	std::vector<WordType> data_;
};

// common virtual surface used by the models (subset of src/Engine/BasisBase.h:41-114)
class BasisBase {
public:
	typedef ProgramGlobals::WordType WordType;
	typedef std::pair<int, int> PairIntType;
	virtual ~BasisBase() { }
	virtual PairIntType parts() const = 0;
	virtual SizeType size() const = 0;
	virtual SizeType dofs() const = 0;
	virtual WordType operator()(SizeType i, SizeType spin) const = 0;
	virtual SizeType perfectIndex(WordType ket1, WordType ket2) const = 0;
	virtual SizeType isThereAnElectronAt(WordType ket1, WordType ket2, SizeType site, SizeType spin, SizeType orb) const = 0;
	virtual SizeType getN(WordType ket1, WordType ket2, SizeType site, SizeType spin, SizeType orb) const = 0;
};

// state i <-> (basis1[i % N_up], basis2[i / N_up]); perfectIndex = rank(up) + rank(down) N_up
// (BasisHubbardLanczos.h:59-63,77-84)
class BasisHubbardLanczos : public BasisBase {
public:
	BasisHubbardLanczos(SizeType nsite, SizeType nup, SizeType ndown) : nup_(nup), ndown_(ndown), basis1_(nsite, nup), basis2_(nsite, ndown) { }
	PairIntType parts() const { return PairIntType(nup_, ndown_); }
	SizeType size() const { return basis1_.size() * basis2_.size(); }
	SizeType dofs() const { return 2; }
	SizeType sizeUp() const { return basis1_.size(); }
	WordType operator()(SizeType i, SizeType spin) const
	{
		return (spin == ProgramGlobals::SPIN_UP) ? basis1_[i % basis1_.size()] : basis2_[i / basis1_.size()];
	}
	SizeType perfectIndex(WordType ket1, WordType ket2) const
	{
		return BasisOneSpin::perfectIndex(ket1) + BasisOneSpin::perfectIndex(ket2) * basis1_.size();
	}
	SizeType isThereAnElectronAt(WordType ket1, WordType ket2, SizeType site, SizeType spin, SizeType) const
	{
		return (((spin == ProgramGlobals::SPIN_UP) ? ket1 : ket2) >> site) & 1;
	}
	SizeType getN(WordType ket1, WordType ket2, SizeType site, SizeType spin, SizeType orb) const
	{
		return isThereAnElectronAt(ket1, ket2, site, spin, orb);
	}

private:
	SizeType nup_, ndown_;
	BasisOneSpin basis1_, basis2_;
};

// All words of nsite digits (bits_ bits each, digit = m + S) with digit sum szPlusConst, ascending (BasisHeisenberg.h:28-46).
// The reference scans every word below 2^(bits*nsite) and keeps the matching ones, and finds a state's index by scanning the
// list (:73-80); here the list is generated digit by digit in the same ascending order and searched by bisection -- same list,
// same indices.  Even spins: digits above twiceS do not occur (mOf, :204-227); odd spins need twiceS + 1 to be a power of two
// for the reference's digit width to hold them.
class BasisHeisenberg : public BasisBase {
public:
	BasisHeisenberg(SizeType nsite, SizeType twiceS, SizeType szPlusConst) : nsite_(nsite), twiceS_(twiceS), szPlusConst_(szPlusConst), bits_(0)
	{
		if (twiceS == 0) err("BasisHeisenberg: HeisenbergTwiceS must be positive\n");
		SizeType lg = 0;
		for (SizeType x = twiceS + 1; x >>= 1;) ++lg; // logBase2, :282-287
		bits_ = 1 + lg;
		if (twiceS & 1) bits_--;
		mask_ = (WordType(1) << bits_) - 1;
		dmax_ = (twiceS & 1) ? SizeType(mask_) : twiceS;
		if (dmax_ < twiceS) err("BasisHeisenberg: the reference's digit width cannot hold this spin\n");
		if (bits_ * nsite > 62) err("BasisHeisenberg: too many sites for a 64-bit word\n");
		fill(0, nsite, szPlusConst);
	}
	PairIntType parts() const { return PairIntType(twiceS_, szPlusConst_); }
	SizeType size() const { return data_.size(); }
	SizeType dofs() const { return twiceS_ + 1; }
	WordType operator()(SizeType i, SizeType) const { return data_[i]; }
	SizeType perfectIndex(WordType ket, WordType) const
	{
		const auto it = std::lower_bound(data_.begin(), data_.end(), ket);
		if (it == data_.end() || *it != ket) throw LppHost::RuntimeError("BasisHeisenberg::perfectIndex: state not in the basis\n");
		return SizeType(it - data_.begin());
	}
	SizeType isThereAnElectronAt(WordType, WordType, SizeType, SizeType, SizeType) const
	{
		throw LppHost::RuntimeError("BasisHeisenberg::isThereAnElectronAt\n");
	}
	SizeType getN(WordType ket1, WordType, SizeType site, SizeType, SizeType) const { return SizeType((ket1 >> (bits_ * site)) & mask_); } // :96-105
	// getBra, :169-193: digit i becomes val1, digit j becomes val2
	WordType getBra(WordType ket, SizeType i, SizeType val1, SizeType j, SizeType val2) const
	{
		WordType bra = ket;
		bra &= ~(mask_ << (i * bits_));
		bra &= ~(mask_ << (j * bits_));
		bra |= WordType(val1) << (i * bits_);
		bra |= WordType(val2) << (j * bits_);
		return bra;
	}
	SizeType szPlusConst() const { return szPlusConst_; }
	SizeType twiceS() const { return twiceS_; }
	SizeType bits() const { return bits_; }

private:
	// most significant digit first, digits ascending: the words come out in ascending order
	void fill(WordType prefix, SizeType left, SizeType sum)
	{
		if (left == 0) {
			if (sum == 0) data_.push_back(prefix);
			return;
		}
		if (sum > dmax_ * left) return;
		for (SizeType d = 0; d <= dmax_ && d <= sum; d++) fill(prefix | (WordType(d) << (bits_ * (left - 1))), left - 1, sum - d);
	}
	SizeType nsite_, twiceS_, szPlusConst_, bits_, dmax_;
	WordType mask_;
	std::vector<WordType> data_;
};

// orbitals == 1: sorted words (down << n) | up with up & down == 0 (BasisTjMultiOrbLanczos.h:29-42,354-369).
// Ascending order = ascending down, then ascending up among the C(n-ndown, nup) ups that avoid `down`,
// so index = rank(down) * C(n-ndown,nup) + rank(up compressed onto the free sites).
class BasisTjMultiOrbLanczos : public BasisBase {
public:
	BasisTjMultiOrbLanczos(SizeType nsite, SizeType nup, SizeType ndown) : n_(nsite), nup_(nup), ndown_(ndown), downs_(nsite, ndown)
	{
		if (nup + ndown > nsite) err("BasisTjMultiOrbLanczos: nup + ndown > sites\n");
		cfree_ = Binomial::get()(nsite - ndown, nup);
		BasisOneSpin ups(nsite - ndown, nup);
		compressedUps_.reserve(ups.size());
		for (SizeType i = 0; i < ups.size(); i++) compressedUps_.push_back(ups[i]);
	}
	PairIntType parts() const { return PairIntType(nup_, ndown_); }
	SizeType size() const { return downs_.size() * cfree_; }
	SizeType dofs() const { return 2; }
	WordType word(SizeType i) const
	{
		const WordType down = downs_[i / cfree_];
		return (down << n_) | deposit(compressedUps_[i % cfree_], ~down & mask());
	}
	WordType operator()(SizeType i, SizeType spin) const
	{
		const WordType w = word(i);
		return (spin == ProgramGlobals::SPIN_UP) ? (w & mask()) : (w >> n_);
	}
	SizeType perfectIndex(WordType ket1, WordType ket2) const
	{
		return BasisOneSpin::perfectIndex(ket2) * cfree_ + BasisOneSpin::perfectIndex(extract(ket1, ~ket2 & mask()));
	}
	SizeType isThereAnElectronAt(WordType ket1, WordType ket2, SizeType site, SizeType spin, SizeType) const
	{
		return (((spin == ProgramGlobals::SPIN_UP) ? ket1 : ket2) >> site) & 1;
	}
	SizeType getN(WordType ket1, WordType ket2, SizeType site, SizeType spin, SizeType orb) const
	{
		return isThereAnElectronAt(ket1, ket2, site, spin, orb);
	}
	// parity of the bits of ket in [i, j)  (doSign, BasisTjMultiOrbLanczos.h:381-400)
	static int doSign(WordType ket, SizeType i, SizeType j)
	{
		const WordType m = ((WordType(1) << j) - 1) & ~((WordType(1) << i) - 1);
		return (__builtin_popcountl(ket & m) & 1) ? -1 : 1;
	}

private:
	WordType mask() const { return (WordType(1) << n_) - 1; }
	static WordType extract(WordType v, WordType m)
	{
		WordType out = 0;
		for (SizeType k = 0; m; k++, m &= m - 1)
			if (v & (m & (~m + 1))) out |= WordType(1) << k;
		return out;
	}
	static WordType deposit(WordType v, WordType m)
	{
		WordType out = 0;
		for (; m && v; v >>= 1, m &= m - 1)
			if (v & 1) out |= m & (~m + 1);
		return out;
	}
	SizeType n_, nup_, ndown_, cfree_;
	BasisOneSpin downs_;
	std::vector<WordType> compressedUps_;
};

} // namespace LanczosPlusPlus
#endif
