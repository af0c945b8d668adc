#!/bin/bash
cd "$(dirname "$0")/.."
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -25
