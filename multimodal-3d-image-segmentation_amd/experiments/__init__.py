"""Training-loop mirror of the reference's experiments package (train_test.py, utils.py) for the
accelerated path; data I/O, CLI, metrics and plotting are out of scope."""
