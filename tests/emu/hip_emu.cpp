// hip_emu.cpp -- fiber scheduler of the wave64 emulator (see hip_emu.h; test infrastructure only).
#include "hip_emu.h"

#include <sys/mman.h>
#include <ucontext.h>

#include <vector>

// Context switch.  glibc's swapcontext / getcontext save and restore the signal mask with one rt_sigprocmask system call
// each: three system calls per emulated GPU thread, which was 40 % of the CPU test-suite's time (sys 4m51 of 11m30).  On
// x86-64 the switch below keeps the callee-saved registers on the fiber's own stack and exchanges stack pointers -- no system
// call; other hosts keep the ucontext path.
#if defined(__x86_64__) && !defined(HIPEMU_USE_UCONTEXT)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* const* load_sp);
asm(R"(
	.text
	.globl hipemu_switch
	.type hipemu_switch,@function
hipemu_switch:
	pushq %rbp
	pushq %rbx
	pushq %r12
	pushq %r13
	pushq %r14
	pushq %r15
	movq %rsp, (%rdi)
	movq (%rsi), %rsp
	popq %r15
	popq %r14
	popq %r13
	popq %r12
	popq %rbx
	popq %rbp
	ret
	.size hipemu_switch, .-hipemu_switch
)");
#endif

dim3 threadIdx, blockIdx, blockDim, gridDim;

namespace hipemu {
namespace {
constexpr size_t STACK_BYTES = 256 * 1024;
constexpr int MAX_THREADS = 1024;
enum State { RUNNABLE, WAIT_BLOCK, WAIT_WAVE, DONE };
struct Fiber {
#ifdef HIPEMU_FAST_SWITCH
	void* sp = nullptr;
#else
	ucontext_t ctx;
#endif
	char* stack = nullptr;
	State state = DONE;
	unsigned gen = 0;  // generation waited on
};
Fiber g_f[MAX_THREADS];
#ifdef HIPEMU_FAST_SWITCH
void* g_sched_sp = nullptr;
#else
ucontext_t g_sched;
#endif
int g_cur = -1, g_nthreads = 0, g_alive = 0;
unsigned g_bar_count = 0, g_bar_gen = 0;
unsigned g_wcount[MAX_THREADS / 64], g_wgen[MAX_THREADS / 64], g_walive[MAX_THREADS / 64];
uint64_t g_slots[MAX_THREADS / 64][64];
const std::function<void()>* g_body = nullptr;

#ifdef HIPEMU_FAST_SWITCH
void yield_to_sched() { hipemu_switch(&g_f[g_cur].sp, &g_sched_sp); }
#else
void yield_to_sched() { swapcontext(&g_f[g_cur].ctx, &g_sched); }
#endif

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
	abort();   // (a finished fiber is never resumed)
}
#ifdef HIPEMU_FAST_SWITCH
// a fresh fiber: six zeroed callee-saved registers under the address hipemu_switch's `ret` jumps to, and above it one slot so
// that fiber_entry starts with the stack alignment of a called function (rsp = 16 k + 8)
void prepare_fiber(Fiber& f)
{
	uintptr_t top = (reinterpret_cast<uintptr_t>(f.stack) + STACK_BYTES) & ~uintptr_t(15);
	void** sp = reinterpret_cast<void**>(top);
	*--sp = nullptr;                                   // (the return address fiber_entry never uses)
	*--sp = reinterpret_cast<void*>(&fiber_entry);     // popped by `ret`
	for (int k = 0; k < 6; k++) *--sp = nullptr;       // rbp rbx r12 r13 r14 r15
	f.sp = sp;
}
#endif
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
#ifdef HIPEMU_FAST_SWITCH
				prepare_fiber(g_f[i]);
#else
				getcontext(&g_f[i].ctx);
				g_f[i].ctx.uc_stack.ss_sp = g_f[i].stack;
				g_f[i].ctx.uc_stack.ss_size = STACK_BYTES;
				g_f[i].ctx.uc_link = nullptr;
				makecontext(&g_f[i].ctx, fiber_entry, 0);
#endif
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
#ifdef HIPEMU_FAST_SWITCH
					hipemu_switch(&g_sched_sp, &f.sp);
#else
					swapcontext(&g_sched, &f.ctx);
#endif
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
