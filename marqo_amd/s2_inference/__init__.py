"""marqo_amd.s2_inference — the reference's s2_inference API surface over the MI355X engine."""
