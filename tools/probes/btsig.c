/* Diagnostic helper: SIGUSR2 makes every thread of the process write its native backtrace to a file descriptor.
 * gcc -shared -fPIC -O1 -o btsig.so btsig.c ; loaded with ctypes by exit_hang_probe.py (never by the product). */
#define _GNU_SOURCE
#include <dirent.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

static int out_fd = 2;
static volatile int fanned_out = 0;

static void handler(int sig, siginfo_t* info, void* uc) {
  (void)uc;
  char line[96];
  const int tid = (int)syscall(SYS_gettid);
  int len = snprintf(line, sizeof line, "\n=== tid %d ===\n", tid);
  if (write(out_fd, line, len) < 0) return;
  void* frames[64];
  const int n = backtrace(frames, 64);
  backtrace_symbols_fd(frames, n, out_fd);
  if (info && info->si_code != SI_TKILL && __sync_bool_compare_and_swap(&fanned_out, 0, 1)) {
    DIR* d = opendir("/proc/self/task");
    if (!d) return;
    struct dirent* e;
    while ((e = readdir(d)) != NULL) {
      const int t = atoi(e->d_name);
      if (t > 0 && t != tid) syscall(SYS_tgkill, getpid(), t, sig);
    }
    closedir(d);
  }
}

int btsig_install(int fd) {
  out_fd = fd;
  void* warm[4];
  backtrace(warm, 4); /* loads libgcc now, not inside the handler */
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_sigaction = handler;
  sa.sa_flags = SA_SIGINFO | SA_RESTART;
  return sigaction(SIGUSR2, &sa, NULL);
}
