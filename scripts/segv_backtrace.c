/* segv_backtrace.c -- diagnostic shim (not part of the product): a SIGSEGV / SIGABRT handler that prints the C backtrace
 * of the faulting thread with backtrace_symbols_fd, then re-raises. Loaded with ctypes by the opt-in child of
 * tests/test_gpu_graph.py to name the frame in which a hipGraph capture that contains an RCCL group dies.
 * build: gcc -O1 -g -shared -fPIC -o segv_backtrace.so segv_backtrace.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig) {
  void*      frames[64];
  const char msg[] = "\n[segv_backtrace] fatal signal, C backtrace of the faulting thread:\n";
  (void)!write(2, msg, sizeof(msg) - 1);
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

void segv_backtrace_install(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof(sa));
  sa.sa_handler = handler;
  sigemptyset(&sa.sa_mask);
  sigaction(SIGSEGV, &sa, 0);
  sigaction(SIGABRT, &sa, 0);
  sigaction(SIGBUS, &sa, 0);
}
