"""Test infrastructure only: CPU restatements of the reference's hot path.
Nothing under sysbio_modeling_amd/ imports this package."""
