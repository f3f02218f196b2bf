// segv_bt.c — a SIGSEGV handler for the GPU box, which has no debugger: LD_PRELOAD it and a crash leaves a native backtrace (and libbivx's
// mapping) in gpurun_out/r4/segv_bt.txt; symbolise with addr2line -f -C -e <a -g build of libbivx.so> <offset>. pytest redirects fd 2, hence the file.
// build: gcc -O1 -g -shared -fPIC -o tools/_segv_bt.so tools/segv_bt.c   (tools/build_variant.sh dbg "-g" <all units> for the library)
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <fcntl.h>
static void handler(int sig, siginfo_t *si, void *ctx) {
  void *bt[64];
  int n = backtrace(bt, 64);
  int fd = open("/root/repo/gpurun_out/r4/segv_bt.txt", O_WRONLY | O_CREAT | O_APPEND, 0644);
  if (fd < 0) fd = 2;
  char msg[128];
  int len = snprintf(msg, sizeof msg, "\n[segv_bt] signal %d at address %p\n", sig, si->si_addr);
  write(fd, msg, len);
  backtrace_symbols_fd(bt, n, fd);
  FILE *f = fopen("/proc/self/maps", "r");
  if (f) { char line[512]; while (fgets(line, sizeof line, f)) if (strstr(line, "libbivx")) write(fd, line, strlen(line)); fclose(f); }
  signal(sig, SIG_DFL);
  raise(sig);
}
__attribute__((constructor)) static void init(void) {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK;
  static char stack[1 << 16];
  stack_t ss = {.ss_sp = stack, .ss_size = sizeof stack, .ss_flags = 0};
  sigaltstack(&ss, 0);
  sigaction(SIGSEGV, &sa, 0);
}
