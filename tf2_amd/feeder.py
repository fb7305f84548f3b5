"""One host thread per HIP stream.

A step is ~40 kernel launches, ~0.2 ms of host time; fed from ONE thread, the fourth of four streams gets its first step
0.6 ms after the first one, and a short run (the driver times 20 steps) pays that skew once at the start and once at the
end.  The library enqueues outside its handle mutex (include/tf2_amd.h, threading note), ctypes releases the GIL for the
call, so one feeder thread per stream puts the steps of all streams on the GPU side by side.

The reference's host has the same shape for the same reason: one command queue per kernel and no blocking call between the
enqueues of a batch (host/src/runner.cpp:85-165)."""
import queue
import threading


class StreamFeeder:
    def __init__(self, streams, runners, device):
        assert len(streams) == len(runners)
        self.device = device
        self._qs = [queue.SimpleQueue() for _ in streams]
        self._errs = []                              # (thread, exception), appended under _lock
        self._lock = threading.Lock()
        self._dead = [False] * len(streams)          # thread i left its loop (error in set-up, or close())
        self._threads = [threading.Thread(target=self._loop, args=(i, st, rn), daemon=True, name=f"tf2-feeder-{i}")
                         for i, (st, rn) in enumerate(zip(streams, runners))]
        for t in self._threads:
            t.start()

    def __len__(self):
        return len(self._qs)

    def _loop(self, i, stream, runner):
        try:
            import torch
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(stream):        # the current stream is per thread
                while True:
                    item = self._qs[i].get()
                    if item is None:
                        return
                    if isinstance(item, threading.Event):
                        item.set()
                        continue
                    try:
                        item(runner)
                    except BaseException as e:     # reported by drain() in the submitting thread
                        with self._lock:
                            self._errs.append((i, e))
        except BaseException as e:                 # set-up failed (set_device / stream): nobody will ever serve this queue
            with self._lock:
                self._errs.append((i, e))
        finally:
            self._dead[i] = True
            while True:                            # release every drain() that is (or will be) waiting on this thread
                try:
                    item = self._qs[i].get_nowait()
                except queue.Empty:
                    break
                if isinstance(item, threading.Event):
                    item.set()

    def submit(self, i, fn):
        """fn(runner) is called on feeder thread i with stream i current; returns at once."""
        self._qs[i].put(fn)

    def drain(self):
        """Returns when everything submitted so far has been ENQUEUED (not executed: synchronise the device for that)."""
        evs = []
        for q in self._qs:
            e = threading.Event()
            q.put(e)
            evs.append(e)
        for i, e in enumerate(evs):
            while not e.wait(0.5):
                if self._dead[i]:                  # the thread died after we queued the event and before it drained its queue
                    break
        self._raise_collected()

    def _raise_collected(self):
        with self._lock:
            errs, self._errs = self._errs, []
        if errs:
            i, first = errs[0]
            if len(errs) > 1:
                raise RuntimeError(f"{len(errs)} errors on feeder threads; first on thread {i}: {first!r}") from first
            raise first

    def close(self):
        for q in self._qs:
            q.put(None)
        for t in self._threads:
            t.join()
        self._raise_collected()                      # errors nobody drained are not lost
