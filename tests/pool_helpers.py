"""A stand-in scheduler for the CPU tests of the GPU worker pool: same client interface as BatchScheduler, no GPU.
Module-level so that worker processes can import it."""
import os
import queue
import threading
import time

import numpy as np


class _Req:
    def __init__(self):
        self.out = queue.Queue()
        self.cancelled = False


class EchoScheduler:
    """Every request yields len(text) chunks of 4 samples: [device, chunk index, voice length, len(text)]."""

    def __init__(self, delay: float = 0.0):
        self.device = os.environ.get("HIP_VISIBLE_DEVICES", "?")
        self.delay = delay
        self.cancels = 0

    def submit(self, text, voice="heart", stream=False, max_new_tokens=None):
        r = _Req()
        if text == "__die__":
            os._exit(3)
        if text == "__cancels__":
            r.out.put(np.array([self.cancels], np.float32))
            r.out.put(None)
            return r
        if text == "__raise__":
            r.out.put(ValueError("bad request"))
            return r

        def run():
            n = len(text) if max_new_tokens is None else min(len(text), max_new_tokens)
            for i in range(n):
                if r.cancelled:
                    break
                r.out.put(np.array([float(self.device), i, len(voice), len(text)], np.float32))
                time.sleep(self.delay)
            r.out.put(None)

        threading.Thread(target=run, daemon=True).start()
        return r

    def iter_chunks(self, r):
        while True:
            item = r.out.get()
            if item is None:
                return
            if isinstance(item, Exception):
                raise item
            yield item

    def cancel(self, r):
        r.cancelled = True
        self.cancels += 1

    def close(self):
        pass


def make_echo(delay: float = 0.0):
    return EchoScheduler(delay)


def make_broken():
    raise RuntimeError("no checkpoint here")
