"""Experiment bookkeeping: the sqlite ModelDatabase."""
