"""Import-path shim: the reference's dotted ``target:`` paths (configs/train.yaml) resolve to the MI355X-native
implementations in ``reface_amd`` so the unchanged YAML + CLI run."""
