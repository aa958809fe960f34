"""stub"""
