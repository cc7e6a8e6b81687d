"""Client/server API surface of the reference's src/algorithms for the contrastive hot path."""
