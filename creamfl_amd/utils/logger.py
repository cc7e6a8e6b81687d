"""Minimal stand-in for src/utils/logger.py:87-139 (PythonLogger): log / report / update_tracker."""
import json
import logging
import sys


class PythonLogger(object):
    def __init__(self, output_file=None, name='creamfl_amd', quiet=False):
        self.logger = logging.getLogger(name)
        self.logger.setLevel(logging.INFO)
        self.logger.propagate = False
        if not self.logger.handlers:
            h = logging.StreamHandler(sys.stderr)
            h.setFormatter(logging.Formatter('%(asctime)s %(message)s'))
            self.logger.addHandler(h)
            if output_file:
                fh = logging.FileHandler(output_file)
                fh.setFormatter(logging.Formatter('%(asctime)s %(message)s'))
                self.logger.addHandler(fh)
        self.quiet = quiet
        self.tracker = {}

    def log(self, msg):
        if not self.quiet:
            self.logger.info(str(msg))

    def report(self, report_dict, prefix='', pretty=False):
        self.log(prefix + (json.dumps(report_dict, indent=2, default=str) if pretty else str(report_dict)))

    def update_tracker(self, data, keys=None):
        for k, v in data.items():
            if keys is None or k in keys:
                self.tracker[k] = v
