"""TEST INFRASTRUCTURE: CPU restatement of the reference (the checker). Never imported by tf2_amd/."""
