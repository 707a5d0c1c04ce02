"""Advisory lock files for the chooser state pickle, NFS-style (atomic symlink
creation), with the same on-disk convention as spearmint/spearmint/Locker.py
(`<file>.lock`), so a reference process and this one exclude each other."""
from __future__ import absolute_import, print_function

import os
import time


class Locker(object):
    def __init__(self):
        self._held = {}

    def __del__(self):
        for name in list(self._held):
            self._held[name] = 1
            self.unlock(name)

    @staticmethod
    def _lockfile(filename):
        return filename + ".lock"

    def lock(self, filename):
        if filename in self._held:
            self._held[filename] += 1
            return True
        try:
            os.symlink(os.devnull, self._lockfile(filename))
        except OSError:
            return False
        self._held[filename] = 1
        return True

    def unlock(self, filename):
        n = self._held.get(filename, 0)
        if n == 0:
            return True
        if n > 1:
            self._held[filename] = n - 1
            return True
        del self._held[filename]
        try:
            os.remove(self._lockfile(filename))
            return True
        except OSError:
            return False

    def lock_wait(self, filename):
        while not self.lock(filename):
            time.sleep(0.01)
