// Stall watchdog (diagnostics): when the host thread stays inside a marked region longer than a threshold, a watchdog thread
// prints what the kernel says that thread is doing (/proc/self/task/<tid>/{stat state, wchan, syscall}) and makes the thread
// print its own native backtrace (SIGUSR2 handler -> backtrace_symbols_fd).  Answers "which runtime call is the host blocked in,
// and is it spinning in user space or sleeping in an ioctl" for stalls that no device profile shows.
//   gcc -O2 -shared -fPIC -o libstallwatch.so stallwatch.c -lpthread
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

static volatile long g_enter_ns = 0;       // 0: outside a region
static volatile int g_dumps = 0;
static pid_t g_tid;
static pthread_t g_main;
static double g_threshold_ms = 15.0;
static int g_fd = 2;

static long now_ns(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1000000000L + ts.tv_nsec;
}

static void on_sig(int sig) {
    void* frames[48];
    int n = backtrace(frames, 48);
    dprintf(g_fd, "stallwatch: native backtrace of the blocked thread (%d frames)\n", n);
    backtrace_symbols_fd(frames, n, g_fd);
}

static void cat_proc(const char* what) {
    char path[128], buf[512];
    snprintf(path, sizeof path, "/proc/self/task/%d/%s", (int)g_tid, what);
    int fd = open(path, O_RDONLY);
    if (fd < 0) { dprintf(g_fd, "stallwatch: %s: unreadable\n", what); return; }
    ssize_t n = read(fd, buf, sizeof buf - 1);
    close(fd);
    if (n < 0) n = 0;
    buf[n] = 0;
    if (strcmp(what, "stat") == 0) {          // keep "pid (comm) STATE"
        char* p = strrchr(buf, ')');
        if (p && p[1]) { p[3] = 0; }
    }
    dprintf(g_fd, "stallwatch: %s: %s%s", what, buf, (n && buf[n - 1] == '\n') ? "" : "\n");
}

static void* watchdog(void* arg) {
    for (;;) {
        usleep(2000);
        long t0 = g_enter_ns;
        if (!t0) continue;
        double ms = (now_ns() - t0) * 1e-6;
        if (ms > g_threshold_ms * (1 + g_dumps) && g_dumps < 4) {
            g_dumps++;
            dprintf(g_fd, "stallwatch: thread %d has been inside the region for %.1f ms\n", (int)g_tid, ms);
            cat_proc("stat");
            cat_proc("wchan");
            cat_proc("syscall");
            cat_proc("stack");
            pthread_kill(g_main, SIGUSR2);
        }
    }
    return 0;
}

int sw_start(double threshold_ms, int fd) {
    void* warm[4];
    backtrace(warm, 4);                       // loads libgcc outside the signal handler
    g_threshold_ms = threshold_ms;
    g_fd = fd;
    g_tid = (pid_t)syscall(SYS_gettid);
    g_main = pthread_self();
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = on_sig;
    sa.sa_flags = SA_RESTART;
    sigaction(SIGUSR2, &sa, 0);
    pthread_t th;
    return pthread_create(&th, 0, watchdog, 0);
}

void sw_enter(void) { g_dumps = 0; g_enter_ns = now_ns(); }
void sw_exit(void) { g_enter_ns = 0; }
