"""Console entry points (reference: samd/inference/)."""
