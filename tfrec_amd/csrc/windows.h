// tfrec_amd/csrc/windows.h -- K1b threshold_kernel (auto threshold) and K2 windows_kernel (trigger windows, work queues).
// One stage of the window-parallel pipeline: included by chains2.hip (inside namespace tfrec, in this order; see the map at its top).
#pragma once

// ------------------------------------------------------------------------------------------------ K1b auto threshold
// fsk_demod::process in auto mode (thresh_mode == 1, fm_demod.cpp:58-73): per block of 8192 decimated samples
//   triggered     = samples at which at least one demodulator is inside its window
//   triggered_avg = (31*triggered_avg + triggered)/32;   every 4th block: avg >= len/32 -> thresh += 2,
//                   avg <= len/64 && thresh > 50 -> thresh -= 2          (len = 16384)
// The threshold of block b+1 depends on block b, so a stream is scanned block by block; all demodulators use
// the same trigger test, so "some demodulator is in its window" = "within Wmax samples after a trigger" with Wmax
// the largest window of the registered demodulators.  One wave per stream: 64 samples per step (coalesced),
// ballot -> mask word, wave-uniform bookkeeping.  Rewrites the trigger mask the front end produced.
__global__ __launch_bounds__(64) void threshold_kernel(const uint32_t *__restrict__ dec, size_t dec_stride,
						       unsigned long long *__restrict__ mask, size_t mask_stride, int n_blocks,
						       FskState *__restrict__ fsk, int wmax)
{
	const int s = blockIdx.x;
	const int lane = threadIdx.x;
	const uint32_t *drow = dec + (size_t)s * dec_stride;
	unsigned long long *mrow = mask + (size_t)s * mask_stride;
	FskState st = fsk[s];
	int last_trig = st.last_trig;  // relative to sample 0 of this submit (very negative: none)
	for (int b = 0; b < n_blocks; b++) {
		int triggered = 0;
		st.runs++;
		for (int w = b * (kBlockDec / 64); w < (b + 1) * (kBlockDec / 64); w++) {
			const uint32_t cw = drow[(w << 6) + lane];
			const int I = (int)(int16_t)(cw & 0xffff), Q = (int)cw >> 16;
			const unsigned long long m = __ballot((abs(I) + abs(Q)) > st.thresh);
			if (lane == 0)
				mrow[w] = m;
			// samples of this word that lie within wmax after the last trigger (windows are >= 355 > 64 long:
			// everything after the word's first trigger is inside)
			const int g0 = w << 6;
			const int first = m ? __builtin_ctzll(m) : 64;
			int carried = last_trig + wmax - g0;  // samples from g0 on still covered by the earlier trigger
			carried = carried < 0 ? 0 : (carried > first ? first : carried);
			triggered += carried + (64 - first);
			if (m)
				last_trig = g0 + 63 - __builtin_clzll(m);
		}
		st.triggered_avg = (31 * st.triggered_avg + triggered) / 32;
		if ((st.runs & 3) == 0) {
			if (st.triggered_avg >= kIndexSpan / 32)
				st.thresh += 2;
			else if (st.triggered_avg <= kIndexSpan / 64 && st.thresh > 50)
				st.thresh -= 2;
		}
	}
	if (lane == 0) {
		const int M = n_blocks * kBlockDec;
		st.last_trig = last_trig - M < -(1 << 28) ? -(1 << 28) : last_trig - M;
		fsk[s] = st;
	}
}

hipError_t launch_threshold(hipStream_t st, const uint32_t *dec, size_t dec_stride, unsigned long long *mask,
			    size_t mask_stride, int n_streams, int n_blocks, FskState *fsk, int wmax)
{
	hipLaunchKernelGGL(threshold_kernel, dim3(n_streams), dim3(64), 0, st, dec, dec_stride, mask, mask_stride, n_blocks, fsk,
			   wmax);
	return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K2
// One WAVE per stream: 64 mask words are loaded coalesced per step, a ballot finds the non-zero ones, and
// a wave-uniform scalar walk over runs of non-zero words maintains, for every active slot of the stream, the
// window state (a window opens at a trigger sample while the timeout counter is 0 and its flush fires W-1
// samples after the last trigger: tfa1.cpp:147-149,179 / tfa2.cpp:351-355,428 / whb.cpp:636-641,691).
// A gap that closes a window is >= W-1 >= 355 samples, so it always spans whole 64-bit words: only the first
// trigger of a run's first word and the last trigger of its last word matter.
__global__ __launch_bounds__(64) void windows_kernel(const unsigned long long *__restrict__ mask, size_t mask_stride,
						     int n_streams, int n_blocks, ChainLaunch L, WinTables T, int long_window)
{
	// The small kernels between the big passes (scan, verify, decode, commit: a few hundred waves of table work) issue
	// ahead of the throughput kernels they share a SIMD with: they cost those nothing measurable and every one of them
	// stands in a stream's chain (-1 % per batch).
	__builtin_amdgcn_s_setprio(3);
	const int s = blockIdx.x;
	const int lane = threadIdx.x;
	const int M = n_blocks * kBlockDec;
	const int nwords = M >> 6;
	const unsigned long long *mrow = mask + (size_t)s * mask_stride;
	const size_t total = (size_t)L.n_active * n_streams * T.cap;
	// per active slot, wave-uniform
	int W[kNSlots], t0[kNSlots], open_g[kNSlots], last_trig[kNSlots], count[kNSlots], vs[kNSlots];
	bool open[kNSlots], overflow = false;
#pragma unroll
	for (int a = 0; a < kNSlots; a++) {
		const bool act = a < L.n_active;
		W[a] = act ? L.params[a].window : 400;
		t0[a] = act ? T.timeout_carry[a * n_streams + s] : 0;
		open[a] = t0[a] > 0;
		open_g[a] = 0;
		last_trig[a] = open[a] ? t0[a] - W[a] : -(1 << 29);  // virtual trigger leaving t0 samples of window
		count[a] = 0;
		vs[a] = 0;
	}
	// work items are collected per wave in LDS and handed to the global queues with ONE atomic per queue and
	// flush (thousands of waves pushing single items contend on a handful of counters otherwise)
	constexpr int kLocal = 256;
	__shared__ uint2 litems[kNQueues][kLocal];
	int lcount[kNQueues];
#pragma unroll
	for (int q = 0; q < kNQueues; q++)
		lcount[q] = 0;
	auto flush_one = [&](int q, int &nq) {  // wave-uniform, q is a compile-time constant at every call site
		if (nq == 0)
			return;
		uint32_t base0 = 0;
		if (lane == 0)
			base0 = atomicAdd(&T.queue[q].count, (uint32_t)nq);
		base0 = __builtin_amdgcn_readfirstlane(base0);
		for (int k = lane; k < nq; k += 64)
			T.items[(size_t)q * total + base0 + k] = litems[q][k];
		nq = 0;
	};
	auto push = [&](int q, uint2 it) {  // wave-uniform; static indexing keeps lcount[] in registers
#pragma unroll
		for (int qq = 0; qq < kNQueues; qq++)
			if (qq == q) {
				if (lcount[qq] == kLocal)
					flush_one(qq, lcount[qq]);
				if (lane == 0)
					litems[qq][lcount[qq]] = it;
				lcount[qq]++;
			}
	};
	auto emit = [&](int a, int og, int close) {
		const int c = a * n_streams + s;
		if (count[a] < T.cap) {
			if (lane == 0) {
				T.open[(size_t)c * T.cap + count[a]] = og;
				T.close[(size_t)c * T.cap + count[a]] = close;
			}
			const int kind = L.params[a].kind;
			const int last = close < M ? close : M - 1;
			const int n = last - og + 1;
			if (kind < 2)  // slicer work item: queues 2*kind + {0 long, 1 short}
				push(2 * kind + (n >= long_window ? 0 : 1), make_uint2((uint32_t)c, (uint32_t)count[a]));
			if (kind == 0 && n >= long_window)  // peak-detector pieces of a long TFA_1 window
				for (int pc = 0; pc * kMarkSlots * 32 < n; pc++)
					push(7, make_uint2((uint32_t)c, (uint32_t)count[a] | ((uint32_t)pc << 17)));
			if (kind > 0) {  // the chain owns a biquad: an item per segment that starts in this window; queue 4: TFA_2
				         // family, 6: WHB
				const int nch = (n + 31) >> 5;
				const int v0 = vs[a];
				for (int v = (v0 + kSegSlots - 1) / kSegSlots * kSegSlots; v < v0 + nch; v += kSegSlots) {
					const int k = v / kSegSlots;
					if (lane == 0)
						T.segstart[(size_t)c * T.segcap + k] = make_uint2((uint32_t)count[a], (uint32_t)(v - v0));
					push(2 + 2 * kind, make_uint2((uint32_t)c, (uint32_t)k));
				}
				vs[a] = v0 + nch;
			}
		} else
			overflow = true;
	};
	for (int w0 = 0; w0 < nwords; w0 += 64) {
		const int w = w0 + lane;
		const unsigned long long m = w < nwords ? mrow[w] : 0ull;
		unsigned long long nz = __ballot(m != 0);
		const int first_bit = m ? __builtin_ctzll(m) : 0;
		const int last_bit = m ? 63 - __builtin_clzll(m) : 0;
		while (nz) {
			const int l = __builtin_ctzll(nz);
			const unsigned long long run = ~(nz >> l);  // its lowest set bit marks where the run of ones from l ends
			const int len = run ? __builtin_ctzll(run) : 64 - l;
			const int l2 = l + len - 1;
			nz = (l2 >= 63) ? 0ull : (nz & (~0ull << (l2 + 1)));
			const int first = ((w0 + l) << 6) + __builtin_amdgcn_readlane(first_bit, l);
			const int lastt = ((w0 + l2) << 6) + __builtin_amdgcn_readlane(last_bit, l2);
#pragma unroll
			for (int a = 0; a < kNSlots; a++) {
				if (a < L.n_active) {
					if (open[a] && first > last_trig[a] + W[a] - 1) {
						emit(a, open_g[a], last_trig[a] + W[a] - 1);
						count[a]++;
						open[a] = false;
					}
					if (!open[a]) {
						open[a] = true;
						open_g[a] = first;
					}
					last_trig[a] = lastt;
				}
			}
		}
	}
#pragma unroll
	for (int a = 0; a < kNSlots; a++) {
		if (a < L.n_active) {
			int tnext = 0;
			if (open[a]) {
				const int close = last_trig[a] + W[a] - 1;
				emit(a, open_g[a], close);
				count[a]++;
				if (close >= M)
					tnext = close - (M - 1);
			}
			if (lane == 0) {
				const int c = a * n_streams + s;
				T.count[c] = count[a] < T.cap ? count[a] : T.cap;
				T.cont[c] = t0[a] > 0 ? 1 : 0;
				T.timeout_next[c] = tnext;
				T.timeout_carry[c] = tnext;
				T.vtotal[c] = vs[a];
			}
		}
	}
#pragma unroll
	for (int q = 0; q < kNQueues; q++)
		flush_one(q, lcount[q]);
	if (overflow && lane == 0)
		*T.overflow = 1;
}
