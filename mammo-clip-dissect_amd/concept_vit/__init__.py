"""Drop-in mirror of the reference's concept_vit/ interface for the dissection hot path."""
