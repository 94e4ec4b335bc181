// Crash diagnostics for the host process (opt-in: aero_install_crash_diagnostics(), or AERO_CRASH_TRACE=1 in the environment when
// the library is loaded). A prover library lives inside somebody else's long-running process (the reference drives its prover
// repeatedly from one worker, aero-sdk/miden-wasm/src/proving_worker.rs:124-223); when that process dies inside native code the
// cause has to reach stderr even if a test harness has redirected file descriptors: the handlers below write with write(2) to a
// duplicate of the stderr that existed at install time AND to the current fd 2.
//   * std::terminate: what() of the escaping exception + native backtrace, then the previous terminate handler;
//   * SIGABRT / SIGSEGV / SIGBUS: signal name, thread id and name, native backtrace (backtrace_symbols_fd is async-signal-safe
//     enough for a process that is about to die; its lines are `object(+0xoffset)`, which llvm-addr2line resolves against the
//     line tables the library is built with), then the previous disposition (so Python's faulthandler and core dumps still happen).
#include <cxxabi.h>
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <typeinfo>

#include "../../include/aero_stark.h"

namespace {

int g_fd_saved = -1;                       // dup of stderr at install time (survives a harness's dup2 over fd 2)
const char* g_fd_path = nullptr;           // AERO_CRASH_LOG: additionally append to this file
std::atomic<bool> g_installed{false};
std::terminate_handler g_prev_terminate = nullptr;
struct sigaction g_prev[3];
const int g_sigs[3] = {SIGABRT, SIGSEGV, SIGBUS};

// fd 2 is written too only when a harness has pointed it somewhere else since install time
bool fd2_differs() {
    if (g_fd_saved < 0) return true;
    struct stat a, b;
    if (fstat(g_fd_saved, &a) != 0 || fstat(2, &b) != 0) return true;
    return a.st_dev != b.st_dev || a.st_ino != b.st_ino;
}
void put(const char* s) {
    size_t n = strlen(s);
    if (g_fd_saved >= 0) (void)!write(g_fd_saved, s, n);
    if (fd2_differs()) (void)!write(2, s, n);
    if (g_fd_path) {
        int fd = open(g_fd_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) { (void)!write(fd, s, n); close(fd); }
    }
}
void put_u(unsigned long v) {
    char b[24]; int i = 23; b[i] = 0;
    do { b[--i] = (char)('0' + v % 10); v /= 10; } while (v && i > 0);
    put(b + i);
}
void put_backtrace() {
    void* frames[64];
    int n = backtrace(frames, 64);
    if (g_fd_saved >= 0) backtrace_symbols_fd(frames, n, g_fd_saved);
    if (fd2_differs()) backtrace_symbols_fd(frames, n, 2);
    if (g_fd_path) {
        int fd = open(g_fd_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
        if (fd >= 0) { backtrace_symbols_fd(frames, n, fd); close(fd); }
    }
}
void put_thread() {
    char name[32] = "?";
    (void)pthread_getname_np(pthread_self(), name, sizeof name);
    put(" tid="); put_u((unsigned long)syscall(SYS_gettid)); put(" pid="); put_u((unsigned long)getpid());
    put(" thread-name="); put(name); put("\n");
}

void on_terminate() {
    put("\n[libaero_stark] std::terminate called");
    put_thread();
    if (std::exception_ptr ep = std::current_exception()) {
        try { std::rethrow_exception(ep); }
        catch (const std::exception& e) { put("[libaero_stark] uncaught exception: "); put(typeid(e).name()); put(": "); put(e.what()); put("\n"); }
        catch (...) { put("[libaero_stark] uncaught exception of a non-std type\n"); }
    } else {
        put("[libaero_stark] no active exception (joinable std::thread destroyed, or terminate called directly)\n");
    }
    put_backtrace();
    if (g_prev_terminate) g_prev_terminate();
    abort();
}

void on_signal(int sig, siginfo_t* info, void* uctx) {
    static std::atomic<int> entered{0};
    if (entered.fetch_add(1) == 0) {
        put("\n[libaero_stark] fatal signal ");
        put(sig == SIGABRT ? "SIGABRT" : sig == SIGSEGV ? "SIGSEGV" : "SIGBUS");
        if (sig != SIGABRT && info) { put(" at address 0x"); char b[20]; snprintf(b, sizeof b, "%lx", (unsigned long)(uintptr_t)info->si_addr); put(b); }
        put_thread();
        put_backtrace();
        put("[libaero_stark] end of native backtrace (the runtime's own message, if any, is above)\n");
    }
    // hand over to whoever was there before (Python's faulthandler, the default action -> core dump)
    for (int i = 0; i < 3; i++) if (g_sigs[i] == sig) {
        const struct sigaction& p = g_prev[i];
        if (p.sa_flags & SA_SIGINFO) { if (p.sa_sigaction) { p.sa_sigaction(sig, info, uctx); } }
        else if (p.sa_handler != SIG_DFL && p.sa_handler != SIG_IGN) { p.sa_handler(sig); }
        sigaction(sig, &p, nullptr);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

}   // namespace

extern "C" int32_t aero_install_crash_diagnostics(void) {
    bool expected = false;
    if (!g_installed.compare_exchange_strong(expected, true)) return AERO_OK;
    g_fd_saved = dup(2);
    if (g_fd_saved >= 0) (void)fcntl(g_fd_saved, F_SETFD, FD_CLOEXEC);
    if (const char* p = getenv("AERO_CRASH_LOG")) if (*p) g_fd_path = strdup(p);
    void* warm[4];
    (void)backtrace(warm, 4);              // loads libgcc now: the first call allocates, which a signal handler should not
    g_prev_terminate = std::set_terminate(on_terminate);
    for (int i = 0; i < 3; i++) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_signal;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER;
        sigemptyset(&sa.sa_mask);
        sigaction(g_sigs[i], &sa, &g_prev[i]);
    }
    return AERO_OK;
}

namespace {
struct AutoInstall {
    AutoInstall() {
        const char* e = getenv("AERO_CRASH_TRACE");
        if (e && e[0] && e[0] != '0') (void)aero_install_crash_diagnostics();
    }
} g_auto_install;
}   // namespace
