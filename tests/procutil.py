"""Bounded child processes for the tests.

Every test that starts a process goes through run_bounded / popen_bounded:
  * the child is a session leader, and on a timeout the WHOLE process group is killed (a host program that forked rank
    processes leaves nobody behind on the GPU);
  * the library's own watchdogs are set well below the test's limit (PGR_EXCHANGE_TIMEOUT_S / ..._COLLECTIVE_TIMEOUT_S),
    so that a stuck rendezvous or collective reports ITSELF (pgr_last_error) before the test has to kill anything;
  * on a timeout the message carries what the child wrote so far and, when a debugger is on the box, the stacks of every
    thread of every process of the group -- the next hang names its blocking call in the test log.
"""
import os
import signal
import subprocess
import time

LIB_TIMEOUTS = {"PGR_EXCHANGE_TIMEOUT_S": "20", "PGR_EXCHANGE_COLLECTIVE_TIMEOUT_S": "30", "HSA_ENABLE_IPC_MODE_LEGACY": "0"}
DEFAULT_LIMIT_S = 90


def child_env(extra=None):
    env = dict(os.environ)
    for k, v in LIB_TIMEOUTS.items():
        env.setdefault(k, v)
    for k, v in (extra or {}).items():
        if v is None:
            env.pop(k, None)  # (None: the child must not see this variable)
        else:
            env[k] = str(v)
    return env


def _group_pids(pgid):
    out = []
    for d in os.listdir("/proc"):
        if not d.isdigit():
            continue
        try:
            if os.getpgid(int(d)) == pgid:
                out.append(int(d))
        except OSError:
            pass
    return sorted(out)


def _stacks(pgid, budget_s=25.0):
    """every thread's stack of every process in the group (rocgdb/gdb if present, else the kernel's view)"""
    dbg = next((p for p in ("/opt/rocm/bin/rocgdb", "/usr/bin/gdb") if os.path.exists(p)), None)
    parts = []
    t0 = time.time()
    for pid in _group_pids(pgid):
        try:
            cmdline = open("/proc/%d/cmdline" % pid, "rb").read().replace(b"\0", b" ").decode(errors="replace")[:200]
        except OSError:
            continue
        parts.append("---- pid %d: %s" % (pid, cmdline))
        left = budget_s - (time.time() - t0)
        got = False
        if dbg and left > 3:
            try:
                r = subprocess.run([dbg, "-p", str(pid), "-batch", "-ex", "set pagination off", "-ex", "thread apply all bt 14"],
                                   capture_output=True, text=True, timeout=left, stdin=subprocess.DEVNULL)
                lines = [l for l in r.stdout.split("\n") if l.startswith("#") or l.startswith("Thread ")]
                if lines:
                    parts.append("\n".join(lines[-160:]))
                    got = True
            except Exception as e:  # the dump is a courtesy
                parts.append("(debugger: %r)" % (e,))
        if not got:
            for tid in sorted(os.listdir("/proc/%d/task" % pid))[:48] if os.path.isdir("/proc/%d/task" % pid) else []:
                try:
                    wchan = open("/proc/%d/task/%s/wchan" % (pid, tid)).read().strip()
                    comm = open("/proc/%d/task/%s/comm" % (pid, tid)).read().strip()
                    parts.append("  tid %s %-16s wchan=%s" % (tid, comm, wchan))
                except OSError:
                    pass
    return "\n".join(parts)


def kill_group(p):
    try:
        os.killpg(p.pid, signal.SIGKILL)
    except OSError:
        pass
    try:
        p.kill()
    except OSError:
        pass


def popen_bounded(cmd, env=None, merge_stderr=True, **kw):
    return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT if merge_stderr else subprocess.PIPE,
                            env=child_env(env), start_new_session=True, stdin=subprocess.DEVNULL, **kw)


class Result:
    def __init__(self, returncode, stdout, stderr, timed_out=False, stacks=""):
        self.returncode, self.stdout, self.stderr, self.timed_out, self.stacks = returncode, stdout, stderr, timed_out, stacks


def communicate_bounded(p, timeout=DEFAULT_LIMIT_S, want_stacks=True):
    """(stdout, stderr, timed_out, stacks) -- never longer than timeout (+ the stack dump), never leaves the group alive"""
    try:
        o, e = p.communicate(timeout=timeout)
        return o, e, False, ""
    except subprocess.TimeoutExpired:
        stacks = _stacks(p.pid) if want_stacks else ""
        kill_group(p)
        try:
            o, e = p.communicate(timeout=10)
        except subprocess.TimeoutExpired:  # a grandchild outside the group still holds the pipe: give up on the rest of the output
            o, e = b"", b""
        return o, e, True, stacks
    finally:
        # a finished host program must not leave rank processes behind either
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass


def run_bounded(cmd, timeout=DEFAULT_LIMIT_S, env=None, check=False, cwd=None):
    """subprocess.run(capture_output=True, text=True) with a hard bound; raises AssertionError with the child's output (and
    stacks) on a timeout, and on a non-zero exit when check is set"""
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=child_env(env), start_new_session=True,
                         stdin=subprocess.DEVNULL, cwd=cwd)
    o, e, timed_out, stacks = communicate_bounded(p, timeout)
    o = (o or b"").decode(errors="replace")
    e = (e or b"").decode(errors="replace")
    if timed_out:
        raise AssertionError("%s\ndid not finish within %d s (process group killed)\n---- stdout tail\n%s\n---- stderr tail\n%s\n---- stacks\n%s"
                             % (" ".join(map(str, cmd)), timeout, o[-2000:], e[-4000:], stacks[-12000:]))
    if check and p.returncode != 0:
        raise AssertionError("%s\nexit code %s\n---- stdout tail\n%s\n---- stderr tail\n%s" % (" ".join(map(str, cmd)), p.returncode, o[-2000:], e[-4000:]))
    return Result(p.returncode, o, e)
