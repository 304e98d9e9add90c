/* segv_backtrace.c -- diagnostic shim (not part of the product): a SIGSEGV / SIGABRT / SIGBUS handler that prints the C
 * backtrace of the faulting thread with backtrace_symbols_fd, then hands the signal to whoever was installed before it
 * (Python's faulthandler) or re-raises it. Loaded with ctypes by the opt-in child of tests/test_gpu_graph.py to name the
 * frame in which a hipGraph capture that contains an RCCL group dies.
 * The handler runs on an ALTERNATE STACK (sigaltstack + SA_ONSTACK): a fault caused by stack exhaustion -- recursion over a
 * captured graph, say -- leaves no room for a handler on the faulting thread's own stack, and the first version of this
 * shim (round 3) therefore printed nothing for exactly the faults it was written for. sigaltstack is per
 * thread: segv_backtrace_install() covers the calling thread; threads the runtime creates afterwards inherit no alternate
 * stack, so a fault there is reported only if their own stack still has room (SA_ONSTACK falls back to it).
 * build: gcc -O1 -g -shared -fPIC -o segv_backtrace.so segv_backtrace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static struct sigaction previous[3];
static const int        signals[3] = {SIGSEGV, SIGABRT, SIGBUS};

static void handler(int sig, siginfo_t* info, void* ctx) {
  void*      frames[64];
  const char msg[] = "\n[segv_backtrace] fatal signal, C backtrace of the faulting thread:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  for (int i = 0; i < 3; i++) {
    if (signals[i] != sig) continue;
    const struct sigaction* p = &previous[i];
    if ((p->sa_flags & SA_SIGINFO) && p->sa_sigaction) {   /* chain: faulthandler prints the Python stacks, then re-raises */
      p->sa_sigaction(sig, info, ctx);
      return;
    }
    if (!(p->sa_flags & SA_SIGINFO) && p->sa_handler != SIG_DFL && p->sa_handler != SIG_IGN) {
      p->sa_handler(sig);
      return;
    }
  }
  signal(sig, SIG_DFL);
  raise(sig);
}

void segv_backtrace_install(void) {
  static char altstack[1 << 16];
  stack_t     ss;
  memset(&ss, 0, sizeof(ss));
  ss.ss_sp   = altstack;
  ss.ss_size = sizeof(altstack);
  (void)sigaltstack(&ss, 0);
  (void)backtrace((void*[1]){0}, 1);   /* loads libgcc's unwinder now: not async-signal-safe on first use */
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_sigaction = handler;
  sa.sa_flags     = SA_SIGINFO | SA_ONSTACK;
  sigemptyset(&sa.sa_mask);
  for (int i = 0; i < 3; i++) sigaction(signals[i], &sa, &previous[i]);
}
