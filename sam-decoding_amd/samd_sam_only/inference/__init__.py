"""Console entry points (reference: samd_sam_only/inference/)."""
