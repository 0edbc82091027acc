// hip_emu.cpp -- fiber scheduler of the wave64 emulator (see hip_emu.h; test infrastructure only).
#include "hip_emu.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
namespace {
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;
enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Fiber {
	ucontext_t ctx;
	char* stack = nullptr;
	State state = DONE;
	unsigned gen = 0;  // generation waited on
};
Fiber g_f[MAX_THREADS];
ucontext_t g_sched;
int g_cur = -1, g_nthreads = 0, g_alive = 0;
unsigned g_bar_count = 0, g_bar_gen = 0;
unsigned g_wcount[MAX_THREADS / 64], g_wgen[MAX_THREADS / 64], g_walive[MAX_THREADS / 64];
uint64_t g_slots[MAX_THREADS / 64][64];
const std::function<void()>* g_body = nullptr;

void yield_to_sched() { swapcontext(&g_f[g_cur].ctx, &g_sched); }

void release_checks_after_exit(int w)
{
	if (g_bar_count > 0 && (int)g_bar_count == g_alive) {
		g_bar_count = 0;
		g_bar_gen++;
	}
	if (g_wcount[w] > 0 && g_wcount[w] == g_walive[w]) {
		g_wcount[w] = 0;
		g_wgen[w]++;
	}
}

void fiber_entry()
{
	(*g_body)();
	const int w = g_cur / 64;
	g_f[g_cur].state = DONE;
	g_alive--;
	g_walive[w]--;
	release_checks_after_exit(w);
	yield_to_sched();
}
}  // namespace

int lane() { return g_cur & 63; }

void syncthreads()
{
	Fiber& f = g_f[g_cur];
	g_bar_count++;
	if ((int)g_bar_count == g_alive) {
		g_bar_count = 0;
		g_bar_gen++;
		return;
	}
	f.gen = g_bar_gen;
	f.state = WAIT_BLOCK;
	yield_to_sched();
}

void wave_sync()
{
	Fiber& f = g_f[g_cur];
	const int w = g_cur / 64;
	g_wcount[w]++;
	if (g_wcount[w] == g_walive[w]) {
		g_wcount[w] = 0;
		g_wgen[w]++;
		return;
	}
	f.gen = g_wgen[w];
	f.state = WAIT_WAVE;
	yield_to_sched();
}

const uint64_t* wave_exchange(uint64_t mine)
{
	const int w = g_cur / 64;
	g_slots[w][g_cur & 63] = mine;
	wave_sync();
	return g_slots[w];
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
	const int nthreads = (int)(block.x * block.y * block.z);
	if (nthreads > MAX_THREADS || nthreads <= 0 || block.y != 1 || block.z != 1) {
		fprintf(stderr, "hipemu: unsupported block shape\n");
		abort();
	}
	for (int i = 0; i < nthreads; i++)
		if (!g_f[i].stack) {
			g_f[i].stack = (char*)mmap(nullptr, STACK_BYTES, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
			if (g_f[i].stack == MAP_FAILED) abort();
		}
	g_body = &body;
	blockDim = block;
	gridDim = grid;
	g_nthreads = nthreads;
	const int nwaves = (nthreads + 63) / 64;
	for (unsigned bz = 0; bz < grid.z; bz++)
	for (unsigned by = 0; by < grid.y; by++)
		for (unsigned bx = 0; bx < grid.x; bx++) {
			blockIdx = dim3(bx, by, bz);
			g_alive = nthreads;
			g_bar_count = 0;
			for (int w = 0; w < nwaves; w++) {
				g_wcount[w] = 0;
				g_walive[w] = (unsigned)((w + 1) * 64 <= nthreads ? 64 : nthreads - w * 64);
				for (int l = 0; l < 64; l++) g_slots[w][l] = 0;  // lanes beyond the block read as zero
			}
			for (int i = 0; i < nthreads; i++) {
				getcontext(&g_f[i].ctx);
				g_f[i].ctx.uc_stack.ss_sp = g_f[i].stack;
				g_f[i].ctx.uc_stack.ss_size = STACK_BYTES;
				g_f[i].ctx.uc_link = nullptr;
				makecontext(&g_f[i].ctx, fiber_entry, 0);
				g_f[i].state = RUNNABLE;
			}
			while (g_alive > 0) {
				bool progress = false;
				for (int i = 0; i < nthreads; i++) {
					Fiber& f = g_f[i];
					if (f.state == DONE) continue;
					if (f.state == WAIT_BLOCK && f.gen == g_bar_gen) continue;
					if (f.state == WAIT_WAVE && f.gen == g_wgen[i / 64]) continue;
					f.state = RUNNABLE;
					g_cur = i;
					threadIdx = dim3((unsigned)i, 0, 0);
					swapcontext(&g_sched, &f.ctx);
					progress = true;
				}
				if (!progress && g_alive > 0) {
					fprintf(stderr,
					        "hipemu: dead-lock in block (%u,%u): a barrier or wave primitive was reached by only part "
					        "of its threads (divergent collective)\n",
					        bx, by);
					abort();
				}
			}
		}
	g_cur = -1;
}
}  // namespace hipemu
