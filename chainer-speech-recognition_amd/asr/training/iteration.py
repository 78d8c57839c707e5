"""Epoch counter of the train scripts (asr/training/iteration.py:22-58): `for epoch in Iteration(n)` announces every epoch and keeps
the wall-clock marks console_log() reports."""
import sys
import time


class Iteration(object):
    def __init__(self, epochs):
        self.epochs = epochs
        self.current_epoch = 0
        self.current_epoch_start_time = 0
        self.start_time = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.current_epoch == self.epochs:
            raise StopIteration()
        now = time.time()
        if self.start_time == 0:
            self.start_time = now
        self.current_epoch += 1
        self.current_epoch_start_time = now
        print("Epoch %d" % self.current_epoch)
        return self.current_epoch

    next = __next__

    def log_progress(self, string):
        sys.stdout.write("\r" + string)
        sys.stdout.flush()

    def console_log(self, d, out=None):
        out = out or sys.stdout
        now = time.time()
        out.write("Epoch {} done in {} min - total {} min\n".format(self.current_epoch, int((now - self.current_epoch_start_time) / 60),
                                                                  int((now - self.start_time) / 60)))

        def walk(values, depth):
            for key in values:
                if isinstance(values[key], dict):
                    out.write("\t" * depth + "%s:\n" % key)
                    walk(values[key], depth + 1)
                else:
                    out.write("\t" * depth + "%s:\t%s\n" % (key, values[key]))
        walk(d, 1)
