/* tests/cpp/segv_trace.c -- test infrastructure (GYMRS_TEST_SEGV_TRACE=1): a SIGSEGV / SIGBUS / SIGABRT handler that prints the NATIVE backtrace of the
 * crashing thread (python's faulthandler shows Python frames only, and names the thread that holds the GIL when a native helper thread crashes) and then
 * hands over to whatever handler was installed before (faulthandler).  gcc -O1 -g -shared -fPIC segv_trace.c -o libsegv_trace.so */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <sys/syscall.h>

static struct sigaction g_prev[64];

static void on_fatal(int sig, siginfo_t* info, void* ctx)
{
    static const char head[] = "\n==== gymrs test: fatal signal, native backtrace of the crashing thread ====\n";
    (void)!write(2, head, sizeof(head) - 1);
    void* frames[64];
    const int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    static const char tail[] = "==== end of native backtrace ====\n";
    (void)!write(2, tail, sizeof(tail) - 1);
    struct sigaction* prev = &g_prev[sig & 63];
    if (prev->sa_flags & SA_SIGINFO) {
        if (prev->sa_sigaction) prev->sa_sigaction(sig, info, ctx);
    } else if (prev->sa_handler != SIG_DFL && prev->sa_handler != SIG_IGN && prev->sa_handler) {
        prev->sa_handler(sig);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

int gymrs_test_install_segv_trace(void)
{
    static char stack[65536];
    stack_t ss;
    memset(&ss, 0, sizeof(ss));
    ss.ss_sp = stack;
    ss.ss_size = sizeof(stack);
    sigaltstack(&ss, 0); /* (the calling thread only; a crash of another thread uses that thread's own stack) */
    const int sigs[] = {SIGSEGV, SIGBUS, SIGABRT, SIGILL, SIGFPE};
    for (unsigned i = 0; i < sizeof(sigs) / sizeof(sigs[0]); ++i) {
        struct sigaction sa;
        memset(&sa, 0, sizeof(sa));
        sa.sa_sigaction = on_fatal;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        if (sigaction(sigs[i], &sa, &g_prev[sigs[i] & 63]) != 0) return -1;
    }
    return 0;
}
