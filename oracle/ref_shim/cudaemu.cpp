/* cudaemu.cpp -- fiber scheduler of oracle/ref_shim/cudaemu.h (test infrastructure only). */
#include "cudaemu.h"

#include <sys/mman.h>
#include <ucontext.h>

uint3 threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace cudaemu {
namespace {
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;
enum State { RUNNABLE, WAITING, DONE };
struct Fiber {
	ucontext_t ctx;
	char* stack = nullptr;
	State state = DONE;
	unsigned gen = 0;
	uint3 tid;
};
Fiber g_f[MAX_THREADS];
ucontext_t g_sched;
int g_cur = -1, g_alive = 0;
unsigned g_bar_count = 0, g_bar_gen = 0;
int g_pred_count = 0, g_pred_result = 0;
const std::function<void()>* g_body = nullptr;

void yield_to_sched() { swapcontext(&g_f[g_cur].ctx, &g_sched); }
void release_barrier()
{
	g_bar_count = 0;
	g_bar_gen++;
	g_pred_result = g_pred_count;
	g_pred_count = 0;
}
void fiber_entry()
{
	(*g_body)();
	g_f[g_cur].state = DONE;
	g_alive--;
	if (g_bar_count > 0 && (int)g_bar_count == g_alive) release_barrier();
	yield_to_sched();
}
}  // namespace

void syncthreads()
{
	Fiber& f = g_f[g_cur];
	g_bar_count++;
	if ((int)g_bar_count == g_alive) {
		release_barrier();
		return;
	}
	f.gen = g_bar_gen;
	f.state = WAITING;
	yield_to_sched();
}

int syncthreads_count(int pred)
{
	if (pred) g_pred_count++;
	syncthreads();
	const int r = g_pred_result;
	syncthreads();   // nobody starts the next count before everybody has read this one
	return r;
}

void launch(dim3 grid, dim3 block, const std::function<void()>& body)
{
	const int nthreads = (int)(block.x * block.y * block.z);
	if (nthreads > MAX_THREADS || nthreads <= 0) {
		fprintf(stderr, "cudaemu: unsupported block shape\n");
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
	for (unsigned bz = 0; bz < grid.z; bz++)
	for (unsigned by = 0; by < grid.y; by++)
	for (unsigned bx = 0; bx < grid.x; bx++) {
		blockIdx = uint3{bx, by, bz};
		g_alive = nthreads;
		g_bar_count = 0;
		g_pred_count = 0;
		for (int i = 0; i < nthreads; i++) {
			getcontext(&g_f[i].ctx);
			g_f[i].ctx.uc_stack.ss_sp = g_f[i].stack;
			g_f[i].ctx.uc_stack.ss_size = STACK_BYTES;
			g_f[i].ctx.uc_link = &g_sched;
			makecontext(&g_f[i].ctx, fiber_entry, 0);
			g_f[i].state = RUNNABLE;
			g_f[i].tid = uint3{(unsigned)i % block.x, ((unsigned)i / block.x) % block.y, (unsigned)i / (block.x * block.y)};
		}
		while (g_alive > 0) {
			bool progressed = false;
			for (int i = 0; i < nthreads; i++) {
				Fiber& f = g_f[i];
				if (f.state == WAITING && f.gen != g_bar_gen) f.state = RUNNABLE;
				if (f.state != RUNNABLE) continue;
				g_cur = i;
				threadIdx = f.tid;
				swapcontext(&g_sched, &f.ctx);
				progressed = true;
			}
			if (!progressed) {
				fprintf(stderr, "cudaemu: deadlock (a barrier was not reached by every live thread of the block)\n");
				abort();
			}
		}
	}
	g_cur = -1;
}

}  // namespace cudaemu
