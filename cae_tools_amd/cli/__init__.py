"""Command-line entry points: train_cae, apply_cae, query_database (python -m cae_tools_amd.cli.<name>)."""
